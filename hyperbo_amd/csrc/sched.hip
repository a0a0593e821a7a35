// Launch schedules: the right-looking blocked Cholesky with look-ahead (panel chain on its own stream, bulk trailing update
// persistent beside it), the block-recursive inverse W = L^-1 walked while the factorisation still runs, and K^-1 = W^T W.
// Everything here decides WHICH kernels of gemm.hip / chol.hip run on WHICH stream in WHAT order; the measurements behind
// the choices are in profiles/r01_potrf_chain.md and profiles/r02_potrf_chain.md.
#include "sched.h"

#include <algorithm>
#include <cmath>

// ---- debug timeline (-DHBO_TIMELINE builds only; tools/timeline.py) ----------------------------------------------------------------
// Every tagged launch gets a pair of device words [first workgroup start, last workgroup end] (100 MHz wall clock) and a host-side
// label; hbo_dbg_timeline(1) starts a recording, hbo_dbg_timeline(0, times, names, ps) ends it and copies everything out.
#ifdef HBO_TIMELINE
namespace {
constexpr int TL_MAX = 16384;
unsigned long long* g_tl_dev = nullptr;
int g_tl_n = 0;
bool g_tl_on = false;
struct TlEntry { char name[16]; int p; };
TlEntry g_tl_entries[TL_MAX];
}
unsigned long long* tl_slot(const char* name, int p) {
  if (!g_tl_on || g_tl_n >= TL_MAX) return nullptr;
  strncpy(g_tl_entries[g_tl_n].name, name, 15); g_tl_entries[g_tl_n].name[15] = 0;
  g_tl_entries[g_tl_n].p = p;
  return g_tl_dev + 2 * g_tl_n++;
}
extern "C" int hbo_dbg_timeline(int begin, unsigned long long* times, char* names, int* ps) {
  if (begin) {
    if (!g_tl_dev && hipMalloc(reinterpret_cast<void**>(&g_tl_dev), sizeof(unsigned long long) * 2 * TL_MAX) != hipSuccess) return -1;
    static unsigned long long init[2 * TL_MAX];
    for (int i = 0; i < TL_MAX; ++i) { init[2 * i] = ~0ull; init[2 * i + 1] = 0; }
    hipMemcpy(g_tl_dev, init, sizeof init, hipMemcpyHostToDevice);
    g_tl_n = 0; g_tl_on = true;
    return 0;
  }
  g_tl_on = false;
  hipDeviceSynchronize();
  if (times) hipMemcpy(times, g_tl_dev, sizeof(unsigned long long) * 2 * g_tl_n, hipMemcpyDeviceToHost);
  for (int i = 0; i < g_tl_n; ++i) { if (names) memcpy(names + 16 * i, g_tl_entries[i].name, 16); if (ps) ps[i] = g_tl_entries[i].p; }
  return g_tl_n;
}
#else
unsigned long long* tl_slot(const char*, int) { return nullptr; }
#endif

// ---- blocked factorisation drivers -----------------------------------------------------------

hipEvent_t pool_event(hbo_ctx* c, size_t i) {
  while (c->ev_pool.size() <= i) {
    hipEvent_t ev;
    hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    c->ev_pool.push_back(ev);
  }
  return c->ev_pool[i];
}


// Matrices of up to this many 128-blocks take 64 x 64 GEMM tiles in the inverse, K^-1 = W^T W and the sweep (option small_nblk; auto:
// fp64 48 -- tools/scan_thresholds.py, profiles/r05_sched_thresholds.md: 40 / 48 at N = 6144 (48 blocks) 5.73 / 5.64 ms, equal below, 64 loses at 56 blocks --
// fp32 32, where the larger matrices' products move to the bf16 cores instead)
static int small_limit(const hbo_ctx* c, int dtype) { return c->opt_small_nblk >= 0 ? c->opt_small_nblk : (dtype == HBO_F64 ? 48 : 32); }
// Batches: the sweep's launches beside the panel chain as plain grids (0), persistent and slot-limited over tiles x tasks from one
// counter (1), also polling the chain's yield table (2).  Auto: persistent up to 8 tasks -- measured with the pipelined cores, ms per
// NLL + gradient, plain / persistent: 4 tasks 1.781 / 1.747, 8 tasks 2.521 / 2.442, 16 tasks 4.200 / 4.223, 32 tasks 7.520 / 7.603,
// 64 tasks 14.13 / 14.31
static int batch_bg(const hbo_ctx* c, int ntasks) { return c->opt_batch_bg >= 0 ? c->opt_batch_bg : (ntasks <= 8 ? 1 : 0); }
// Right-looking blocked Cholesky with look-ahead.  Panels are 128 wide; `group` consecutive panels
// are factored left-looking (the later ones first receive the group's earlier panels: syrk_col),
// then one trailing update with K = 128*group is applied.  The trailing update is split in two
// launches: F1 updates only the NEXT group's block columns, F2 the rest; the next group's panel
// work (potf2 -> trsm, the serial chain) runs on a second stream as soon as F1 is done, so F2
// -- the bulk of the flops -- overlaps it.
void run_potrf(hbo_ctx* c, int dtype, const TaskDesc* d_tasks, int ntasks, int max_nblk, int* d_info, TrtriProgress* early,
               SweepState* sweep) {
  // panels per trailing update and CUs the persistent bulk update leaves to the panel chain.  Measured (NLL+grad, ms):
  //   N = 4096: (4, 32) 3.61, (3, 32) 3.54, (3, 64) 3.51;   N = 8192: (4, 32) 13.49, (3, 32) 13.34, (3, 64) 13.25,
  //   (3, 96) 13.43, (2, 64) 13.57, (5, 32) 13.61;   N = 16384: (4, 32) 78.4, (3, 32) 79.1, (3, 64) 79.5
  const bool small_mat = max_nblk <= 96;
  //   round 2, N = 65536: group 4 / 6 / 8 / 12 / 16: 60.2 / 60.7 / 62.0 / 62.6 / 62.0 TFLOP/s; N = 32768: 4 / 6 / 8: 56.8 / 57.2 / 57.9; N = 16384: 48.1 / 47.4 / 47.8
  //   round 3, fp32 with the updates on the bf16 cores, N = 16384 (factor, ms): group 4 / 5 / 6 / 7 / 8: 23.1 / 22.5 / 22.3 / 22.4 / 22.0
  const bool s3_large = dtype == HBO_F32 && c->opt_syrk_bf16x3 && !small_mat;
  // (without look-ahead -- small problems, see use_lookahead -- there is no chain / bulk distinction to group for: one panel per
  //  trailing update, the plain right-looking form: N = 512 / 1024 / 2048: 0.399 -> 0.383, 0.686 -> 0.654, 1.33 -> 1.28 ms; up to
  //  eight tasks -- 64 tasks of 4 blocks: 0.761 with groups of three, 0.796 with one)
  const bool la_on = use_lookahead(c, ntasks, max_nblk);
  //   round 5 (tools/ab_suite.py; fp64 one matrix, group 3 / 5 / 6 / 7 / 8): N = 8192 10.57 / 10.54 / 10.58 / 10.56 / 10.56 (flat), 12288 31.98 / 31.21 / 31.16 /
  //   30.97 / 31.02; fp32 with the trailing updates on the fp16 cores (they are short: the chain and the launch count decide): N = 8192 group 3 / 6 / 8 / 12
  //   6.09 / 5.82-5.87 / 5.92 / 6.16-6.19; N = 16384 (factor, ms) 8 / 12 / 16 18.5 / 18.3-18.5 / 18.7-19.0 and, with two-level groups (outer / inner: see
  //   group_inner below), 16 / 8 18.0-18.2, 12 / 6 18.2-18.5, 8 / 4 18.6
  const bool s3_one = dtype == HBO_F32 && c->opt_syrk_bf16x3 && ntasks == 1;
  const int q_small = (s3_one && max_nblk >= 56) ? 6 : ((ntasks == 1 && max_nblk > 64) ? 7 : 3);
  const int q = c->opt_group > 0 ? c->opt_group : ((!la_on && ntasks <= 8) ? 1 : (small_mat ? q_small : (s3_large ? 16 : (max_nblk >= 256 ? 8 : 4))));
  const int q_inner = c->opt_group_inner >= 0 ? c->opt_group_inner : ((s3_large && q == 16) ? 8 : 0);
  //   with the CU yield (below): N = 8192 (3, 64) 12.58, (3, 48) 12.52, (3, 32) 12.61, (3, 16) 13.24, (4, 48) 12.66
  //   round 2 (chain kernels mark their CUs, background workgroups there pause): 32 beats 48 at N = 8192 (11.73 / 11.81)
  const int persist_free = c->opt_persist_free >= 0 ? c->opt_persist_free : 32;
  hipStream_t sm = c->stream;
  // (a single block column has no trailing matrix to look ahead over: it stays on the caller's stream)
  const bool la = la_on;
  hipStream_t sp = la ? c->stream2 : c->stream;
  hipStream_t sb = sm;   // bulk updates share the main stream (CU-masked queues were measured slower)
  size_t evi = 0;
  if (la) { hipEvent_t e = pool_event(c, evi++); hipEventRecord(e, sm); hipStreamWaitEvent(sp, e, 0); }
  hipEvent_t ev_f1 = nullptr, ev_f2 = nullptr, ev_f1b = nullptr;
  // (measured: batches gain -- 64 tasks 14.32 -> 14.12 ms, the 8-task shard 2.60 -> 2.53; one matrix does not -- N = 8192
  //  factorisation 5.35 -> 5.41, N = 4096 2.63 -> 2.68: the two cross-stream hops cost what the shorter launch saves)
  const bool split_f1 = use_lookahead(c, ntasks, max_nblk) && c->stream3 && (c->opt_split_f1 >= 2 || (c->opt_split_f1 == 1 && ntasks > 1));
  // early inverse: a batch takes every piece as soon as four more panels are final (16.9 against 17.35 ms for 64 tasks
  // of ~2000 points, 3.77 against 3.92 for 8); one large matrix only at half time -- more launches on the side stream
  // take slots from the panel chain (N = 8192: 13.63 ms at 32 panels, 13.8 at 4, 14.0 at 2)
  int tgran = 0;
  // (one large matrix, measured later with the CU yield below: a single call after 13/16 of the panels instead of half --
  //  N = 8192, call after panel 32 / 40 / 44 / 48 / 52 / 56 / 60: 12.42 / 12.37 / 12.34 / 12.25 / 12.21 / 12.30 / 12.50 ms)
  int early_at = -1;   // single-task form: the one panel count after which the side stream gets its work
  if (tgran <= 0) {
    tgran = 4;
    if (ntasks == 1) {
      // (round 2, with the persistent / yielding forms of the co-running products -- they no longer stall the chain --
      //  N = 8192, (start after panel, CUs left free by the inverse, by the bulk update): (40, 48, 32) 11.73 ms,
      //  (36, 64, 32) 11.74, (40, 32, 32) 11.80, (44, 48, 32) 11.92, (52, 16, 48) 12.27, (48, 96, 32) 12.37)
      early_at = c->opt_trtri_at > 0 ? std::min(c->opt_trtri_at * max_nblk / 64, max_nblk - 1) : (max_nblk * 5 / 8) & ~3;
      if (early_at < 4) early_at = -1;
      tgran = 1 << 30;
    }
  }
  // (up to 96 blocks: N = 4096 3.37 -> 3.29 ms, N = 8192 12.61 -> 12.52; N = 16384 loses 0.9 % to the polling)
  // fp32 with the trailing updates on the bf16 cores: the bulk update is 1.5x shorter and the panel chain sets the pace at
  // every size, so the chain's kernels are protected as for the small matrices
  const bool s3_wanted = dtype == HBO_F32 && c->opt_syrk_bf16x3 && max_nblk > 1;
  const bool batch_yield = la && ntasks > 1 && sweep && batch_bg(c, ntasks) >= 2 && c->opt_cu_yield;
  int* const yield_flag = ((la && ntasks == 1 && c->opt_cu_yield && (small_mat || s3_wanted)) || batch_yield) ? c->d_yield : nullptr;
  if (yield_flag) hipMemsetAsync(yield_flag, 0, sizeof(int) * HBO_YIELD_TAB_ENTRIES, sm);
  int* const chain_mark = (yield_flag && c->opt_cu_yield >= 2) ? yield_flag : nullptr;   // the chain's wide kernels mark their CUs too
  c->gemm_yield = yield_flag;
  // one tile counter per bulk launch (dynamic tile assignment of the persistent form), zeroed up front
  int* counters = (int*)ws_get(c, WS_COUNTERS, sizeof(int) * HBO_N_COUNTERS);
  int n_counter = 0;
  if (counters) hipMemsetAsync(counters, 0, sizeof(int) * HBO_N_COUNTERS, sm);
  c->trtri_counters = counters ? counters + HBO_N_BULK_COUNTERS : nullptr;   // the rest: the persistent inverse products (trtri_level, sweep_advance)
  c->trtri_counter_next = 0;
  c->post_counter_next = 0;   // (post_counter: the launches behind this factorisation)
  // fp32: every trailing update on the bf16 matrix cores from an exact three-way split of the group's panels (post3.hip:
  // syrk3_kernel; 1.4x the fp32-MFMA rate at fp32 accuracy).  Each panel is split right behind its solve; two buffers alternate
  // by group, because the bulk update of group g still reads its panels while group g + 1 is being solved.
  const bool s3 = dtype == HBO_F32 && c->opt_syrk_bf16x3 && max_nblk > 1;
  Syrk3Args s3a = {};
  unsigned short* s3buf[2] = {nullptr, nullptr};
  if (s3) {
    s3a.tasks = d_tasks; s3a.nkb = q * (HBO_TILE / 16);
    s3a.task_stride = (int64_t)(max_nblk + 1) * s3a.nkb * 3 * (HBO_TILE * 16);
    const size_t bytes = sizeof(unsigned short) * (size_t)s3a.task_stride * ntasks;
    s3buf[0] = static_cast<unsigned short*>(ws_get(c, WS_SYRK3_A, bytes));
    s3buf[1] = static_cast<unsigned short*>(ws_get(c, WS_SYRK3_B, bytes));
  }
  const bool use_s3 = s3 && s3buf[0] && s3buf[1];
  // ... or, when the caller knows the largest diagonal entry (|L_ij| <= sqrt(max A_ii) gives the panels' scale without a pass), on
  // the fp16 cores from a TWO-way split: three MFMAs per product instead of six (Syrk3Args::h2; the augmented tile-row carries
  // measured scales per 16 rows x 64 columns, one array per panel buffer)
  float* h2_aug = nullptr;
  c->h2_words = nullptr;
  if (dtype == HBO_F32 && c->opt_chol_f16x2 && c->chol_diag_bound > 0) {
    // measured maxima of the inverse's operands (trtri_level3, run_lauum), zeroed once per factorisation
    c->h2_words = static_cast<unsigned int*>(ws_get(c, WS_H2_SCALES, sizeof(unsigned int) * HBO_H2_WORDS));
    if (c->h2_words) hipMemsetAsync(c->h2_words, 0, sizeof(unsigned int) * HBO_H2_WORDS, sm);
  }
  if (use_s3 && c->h2_words) {
    s3a.aug_stride = (int64_t)(s3a.nkb / 4) * 8;
    h2_aug = static_cast<float*>(ws_get(c, WS_H2_AUG, sizeof(float) * 2 * (size_t)s3a.aug_stride * ntasks));
    if (h2_aug) { s3a.h2 = 1; s3a.sx = s3a.sy = post2h_scale_for(std::sqrt(c->chol_diag_bound)); }
  }
  auto tiles_of = [&](int c_lo, int c_hi) { int n = 0; for (int cc = c_lo; cc < std::min(c_hi, max_nblk); ++cc) n += max_nblk + 1 - cc; return n; };
  int grp_index = 0;
  for (int g0 = 0; g0 < max_nblk; g0 += q, ++grp_index) {
    const int g1 = std::min(g0 + q, max_nblk);
    const int g2 = std::min(g1 + q, max_nblk);
    if (use_s3) s3a.Xp = s3buf[grp_index & 1];
    if (s3a.h2) s3a.aug_scale = h2_aug + (grp_index & 1) * s3a.aug_stride * ntasks;
    if (la && ev_f1) hipStreamWaitEvent(sp, ev_f1, 0);
    // two-level groups (group_inner = qi, 0 < qi < q): the chain's column updates stay short -- inside the inner group [gi, gi + qi) --
    // and at every inner boundary ONE update brings the group's remaining columns up to date with the inner group just finished
    // (K = 128 qi, on the chain); the trailing updates F1 / F2 keep the outer group's K = 128 q
    const int qi = (q_inner > 0 && q_inner < q && !c->opt_syrk3_col) ? q_inner : 0;
    for (int p = g0; p < g1; ++p) {
      const int gi = qi ? g0 + (p - g0) / qi * qi : g0;   // first panel of p's inner group
      if (qi && p == gi && p > g0) {
        if (use_s3) {   // the inner group just finished, as split planes (its rows below; blocks [gi - qi - g0 ..) of the buffer)
          ProfScope ps(c, "split3", 2, sp);
          Syrk3Args a = s3a; a.kcol0 = (gi - qi) * HBO_TILE; a.nk_split = qi * (HBO_TILE / 16); a.r_lo = gi; a.kb_off = (gi - qi - g0) * (HBO_TILE / 16);
          launch_split3_panel(a, max_nblk + 1 - gi, ntasks, sp);
        }
        if (ev_f1b) { hipStreamWaitEvent(sp, ev_f1b, 0); ev_f1b = nullptr; }
        ProfScope ps(c, "syrk_inner", 2, sp);
        if (use_s3) {
          Syrk3Args a = s3a; a.kb_off = (gi - qi - g0) * (HBO_TILE / 16); a.nk = qi * (HBO_TILE / 16); a.c_lo = gi; a.c_hi = g1;
          a.yield_mark = chain_mark;
          launch_syrk3(a, tiles_of(gi, g1), ntasks, sp);
        } else {
          GemmArgs a = {}; a.tasks = d_tasks; a.mode = GEMM_SYRK; a.p0 = gi - qi; a.kt = qi; a.c_lo = gi; a.c_hi = g1; a.aug = 1;
          a.small_tiles = (int64_t)(max_nblk + 1 - gi) * (g1 - gi) * ntasks < 600;
          a.yield_mark = chain_mark; a.tl = tl_slot("syrk_inner", p);
          launch_gemm(dtype, a, dim3(max_nblk + 1 - gi, g1 - gi, ntasks), sp);
        }
      }
      if (p > g0 && use_s3 && c->opt_syrk3_col) {
        ProfScope ps(c, "syrk_col", 2, sp);
        Syrk3Args a = s3a; a.kb_off = 0; a.nk = (p - g0) * (HBO_TILE / 16); a.c_lo = p; a.c_hi = p + 1;
        a.yield_mark = chain_mark;
        launch_syrk3(a, tiles_of(p, p + 1), ntasks, sp);
      } else if (p > gi) {  // left-looking update of block column p with the (inner) group's earlier panels
        if (ev_f1b) { hipStreamWaitEvent(sp, ev_f1b, 0); ev_f1b = nullptr; }   // (the previous group's contribution to this column)
        ProfScope ps(c, "syrk_col", 2, sp);
        GemmArgs a = {}; a.tasks = d_tasks; a.mode = GEMM_SYRK; a.p0 = gi; a.kt = p - gi; a.c_lo = p; a.c_hi = p + 1; a.aug = 1; a.small_tiles = 1;
        a.yield_mark = chain_mark; a.tl = tl_slot("syrk_col", p);
        launch_gemm(dtype, a, dim3(max_nblk + 1 - p, 1, ntasks), sp);
      }
      { ProfScope ps(c, "potf2", 2, sp); launch_potf2(dtype, d_tasks, ntasks, p, d_info, sp, yield_flag, tl_slot("potf2", p)); }
      {
        // (fp32 + bf16x3 updates: the solve writes its panel as three bf16 planes too -- no separate split launch on the chain)
        SplitOut so = {};
        if (use_s3) { so.xp = s3a.Xp; so.task_stride = s3a.task_stride; so.nkb = s3a.nkb; so.kb_off = (p - g0) * (HBO_TILE / 16); }
        const bool fused = use_s3 && !c->opt_syrk3_sep && c->opt_syrk3_col && !s3a.h2;   // (the f16x2 split needs the augmented rows' maxima first: a kernel of its own)
        { ProfScope ps(c, "trsm", 2, sp); launch_trsm(dtype, d_tasks, ntasks, p, max_nblk, sp, chain_mark, fused ? &so : nullptr, tl_slot("trsm", p)); }
        if (use_s3 && !fused && p + 1 < max_nblk && (c->opt_syrk3_col || p + 1 == g1)) {
          // the column updates inside the group stay on fp32 MFMA (64x64 tiles): ONE split of the whole group behind its last
          // solve, for the wide updates (F1, F2) -- or, with syrk3_col, one per panel for the column updates too
          ProfScope ps(c, "split3", 2, sp);
          const int pfirst = c->opt_syrk3_col ? p : gi;   // (two-level groups: the earlier inner groups were split at their boundaries)
          Syrk3Args a = s3a; a.kcol0 = pfirst * HBO_TILE; a.nk_split = (p + 1 - pfirst) * (HBO_TILE / 16); a.r_lo = p + 1; a.kb_off = (pfirst - g0) * (HBO_TILE / 16);
          launch_split3_panel(a, max_nblk + 1 - (p + 1), ntasks, sp);
        }
      }
      if (sweep && (p + 1) % sweep->qs == 0 && p + 1 < max_nblk) {
        // block columns 0..p of L are final: the row group that ends here goes through the one-sweep inverse on the side stream
        hipEvent_t e = pool_event(c, evi++);
        hipEventRecord(e, sp);
        hipStreamWaitEvent(c->stream4, e, 0);
        ProfScope ps(c, "sweep_early", 1, c->stream4);
        sweep_advance(c, dtype, d_tasks, ntasks, max_nblk, p + 1, c->stream4, *sweep);
      } else if (early && ((p + 1) % tgran == 0 || p + 1 == early_at) && p + 1 < max_nblk) {
        // block columns 0..p of L are final: everything of the inverse that only needs them goes to a side stream
        // (the panel chain leaves most of the machine idle in the second half of the factorisation)
        hipEvent_t e = pool_event(c, evi++);
        hipEventRecord(e, sp);
        hipStreamWaitEvent(c->stream4, e, 0);
        ProfScope ps(c, "trtri_early", 1, c->stream4);
        trtri_advance(c, dtype, d_tasks, ntasks, max_nblk, p + 1, c->stream4, *early);
      }
    }
    // F1 (next group's block columns) is on the critical path: with look-ahead it is launched on the panel stream
    // itself -- no cross-stream event hop before and after it -- once the previous bulk update, which wrote the same
    // tiles, is done (ev_f2); the main stream only learns that F1 is finished (ev_f1) to start F2 behind it.
    hipStream_t s1 = la ? sp : sm;
    if (la && s1 == sm) { hipEvent_t e = pool_event(c, evi++); hipEventRecord(e, sp); hipStreamWaitEvent(sm, e, 0); }
    if (g1 < max_nblk) {
      GemmArgs a = {}; a.tasks = d_tasks; a.mode = GEMM_SYRK; a.p0 = g0; a.kt = g1 - g0; a.aug = 1;
      {
        // F1 is a chain kernel when it runs on the panel stream: it marks its CUs instead of polling
        a.yield_flag = (la && chain_mark) ? nullptr : yield_flag;
        a.yield_mark = la ? chain_mark : nullptr;
        if (s1 == sp && ev_f2) hipStreamWaitEvent(sp, ev_f2, 0);
        if (ev_f1b) { hipStreamWaitEvent(sp, ev_f1b, 0); ev_f1b = nullptr; }   // (a group of one panel never waited for it)
        ProfScope ps(c, "syrk_trailing", 1, s1);
        a.c_lo = g1; a.c_hi = la ? g2 : max_nblk;
        if (use_s3) {
          Syrk3Args b = s3a; b.kb_off = 0; b.nk = (g1 - g0) * (HBO_TILE / 16); b.c_lo = a.c_lo; b.c_hi = a.c_hi;
          b.yield_mark = a.yield_mark; b.yield_flag = a.yield_flag;
          launch_syrk3(b, tiles_of(b.c_lo, b.c_hi), ntasks, s1);
        } else {
        if (split_f1 && s1 == sp && a.c_hi - a.c_lo > 1) {
          // only the NEXT block column is on the critical path (potf2 and the solve of panel g1 read nothing else): the group's later
          // columns go to a third stream and are waited for by the first column update that touches them
          hipEvent_t e = pool_event(c, evi++);
          hipEventRecord(e, sp); hipStreamWaitEvent(c->stream3, e, 0);
          GemmArgs b = a; b.c_lo = a.c_lo + 1;
          b.small_tiles = (int64_t)(max_nblk + 1 - b.c_lo) * (b.c_hi - b.c_lo) * ntasks < 600;
          b.tl = tl_slot("f1b", g1);
          launch_gemm(dtype, b, dim3(max_nblk + 1 - b.c_lo, b.c_hi - b.c_lo, ntasks), c->stream3);
          ev_f1b = pool_event(c, evi++);
          hipEventRecord(ev_f1b, c->stream3);
          a.c_hi = a.c_lo + 1;
        }
        // few tiles (one group's block columns, or a small remainder): 64x64 tiles for latency
        a.small_tiles = (int64_t)(max_nblk + 1 - a.c_lo) * (a.c_hi - a.c_lo) * ntasks < 600;
        a.tl = tl_slot("f1", g1);
        launch_gemm(dtype, a, dim3(max_nblk + 1 - a.c_lo, a.c_hi - a.c_lo, ntasks), s1);
        a.tl = nullptr;
        }
      }
      if (la) {
        ev_f1 = pool_event(c, evi++);
        hipEventRecord(ev_f1, s1);
        if (s1 == sp) { hipStreamWaitEvent(sm, ev_f1, 0); ev_f1 = nullptr; }   // the panel stream continues in order
        if (g2 < max_nblk) {
          // F2 on the CU-masked bulk stream: after F1(g) (same C columns are not shared, but F2(g)
          // must precede F1(g+1)/F2(g+1) which accumulate into the same tiles)
          if (ev_f1) hipStreamWaitEvent(sb, ev_f1, 0);
          {
            a.yield_flag = yield_flag; a.yield_mark = nullptr;   // the bulk update is background work
            a.c_lo = g2; a.c_hi = max_nblk;
            const int64_t m = max_nblk - g2;
            a.small_tiles = m * (m + 1) / 2 * ntasks < 600;
            if (use_s3) {
              ProfScope ps(c, a.small_tiles ? "syrk_trailing" : "syrk_bulk", 1, sb);
              Syrk3Args b = s3a; b.kb_off = 0; b.nk = (g1 - g0) * (HBO_TILE / 16); b.c_lo = g2; b.c_hi = max_nblk;
              // persistent, two workgroups on all but `s3_free` CUs: the panel kernels beside it always find a CU with room
              const int nt = tiles_of(g2, max_nblk);
              const int pb = 2 * (c->n_cus - c->opt_syrk3_free);
              if (ntasks == 1 && la && c->opt_syrk3_free > 0 && nt > pb && counters && n_counter < HBO_N_BULK_COUNTERS) { b.persistent = pb; b.work_counter = counters + n_counter++; }
              b.yield_flag = yield_flag;
              launch_syrk3(b, nt, ntasks, sb);
            } else {
            // "syrk_bulk" = the 128x128-tile bulk trailing update (the roofline kernel of bench.py)
            // (round 6: leaving the chain 16 CUs while the trailing matrix has >= 48 tile columns and 48-64 afterwards -- early groups wait for the bulk
            //  update, later ones for the chain -- measured neutral: N = 8192 10.64 -> 10.57-10.75 ms, N = 6144 5.60 -> 5.54-5.60; removed)
            const int pblocks = 2 * (c->n_cus - persist_free);
            // Two launches (hbo_tune f2_split): the NEXT F1 accumulates into block columns [g2, g3) only, which this update writes FIRST
            // (column-major tile order) -- but an event fires at the end of a launch, so the chain's next F1 waited for the whole bulk
            // update and the bulk update for F1: F1 -> hop -> F2 -> hop per group (profiles/r05_chain_timeline.md).  With the leading
            // columns -- at least the next group's, and about one resident round of tiles -- as a launch of their own and the event
            // behind THAT, the chain runs up to one bulk launch ahead and neither stream waits for the other at every group.
            // MEASURED NEUTRAL (round 6, profiles/r06_f2_split.md): identical values, N = 8192 10.51 -> 10.57-10.59 ms, N = 6144 5.55 -> 5.58, shard 2.412 ->
            // 2.410, 64 tasks 13.95 -> 13.94.  The timeline shows why: the two launches take 457 us where the one took 414 (two ramps, two
            // partly filled last rounds), which is what the shorter idle gap of the bulk stream (75 -> 49 us per group) gives back.  Off by default.
            int c_split = max_nblk;
            if (c->opt_f2_split && (ntasks == 1 || c->opt_f2_split >= 2) && !a.small_tiles) {
              const int g3 = std::min(g2 + q, max_nblk);
              int cs = g3;
              while (cs < max_nblk && tiles_of(g2, cs) < pblocks) ++cs;
              // (what is left must be worth a launch: at least half a resident round, or the update stays whole)
              if (cs < max_nblk && tiles_of(cs, max_nblk) * 2 >= pblocks) c_split = cs;
            }
            for (int part = 0; part < 2; ++part) {
              a.c_lo = part == 0 ? g2 : c_split; a.c_hi = part == 0 ? c_split : max_nblk;
              if (a.c_lo >= a.c_hi) continue;
              {
              ProfScope ps(c, a.small_tiles ? "syrk_trailing" : "syrk_bulk", 1, sb);
              // persistent form (single task, enough tiles to fill the machine): leave CUs for the panel chain
              const int64_t ntiles = (int64_t)tiles_of(a.c_lo, a.c_hi) * (a.small_tiles ? 4 : 1);
              // (for large trailing matrices the bulk update dominates and gets the whole machine)
              a.persistent = (ntasks == 1 && persist_free > 0 && ntiles > pblocks && m <= 96) ? pblocks : 0;
              // (beyond that the whole machine, but still as a resident grid drawing tiles from the counter: see launch_gemm_t, LAUUM)
              if (!a.persistent && ntasks == 1 && persist_free > 0 && m > 96 && c->opt_lauum_persist && !a.small_tiles) a.persistent = 2 * c->n_cus;
              a.work_counter = (a.persistent && counters && n_counter < HBO_N_BULK_COUNTERS) ? counters + n_counter++ : nullptr;
              a.n_big = 0;
              if (a.persistent && a.work_counter && !a.small_tiles) {
                // a partly filled last round (fewer than half of the workgroups would get a 128-tile) runs on 64-tiles
                // (the whole last round on 64-tiles, or never: measured equal or slower, profiles/r02_potrf_chain.md)
                const int64_t rem = ntiles % pblocks;
                if (rem > 0 && rem * 2 <= pblocks) a.n_big = (int)(ntiles - rem);
              }
              a.tl = tl_slot(part == 0 ? "f2" : "f2b", g1);
              launch_gemm(dtype, a, dim3(max_nblk + 1 - a.c_lo, a.c_hi - a.c_lo, ntasks), sb);
              a.persistent = 0; a.work_counter = nullptr; a.n_big = 0; a.tl = nullptr;
              }
              if (part == 0) {   // what the next F1 (and nothing else on the chain) waits for
                hipEvent_t e2 = pool_event(c, evi++);
                hipEventRecord(e2, sb);
                ev_f2 = e2;
              }
            }
            }
          }
          if (use_s3) {
            hipEvent_t e2 = pool_event(c, evi++);
            hipEventRecord(e2, sb);
            ev_f2 = e2;
          }
        }
      }
    }
  }
  c->gemm_yield = nullptr;
  c->trtri_counters = nullptr;
  if (ev_f1b) { hipStreamWaitEvent(sp, ev_f1b, 0); ev_f1b = nullptr; }
  if (la) { hipEvent_t e = pool_event(c, evi++); hipEventRecord(e, sp); hipStreamWaitEvent(sm, e, 0); }   // join
  // (the sweep's tail, on the main stream, waits for exactly what it needs of the side stream's work: sweep_advance)
  if (early && !sweep) {   // the rest of the inverse (main stream) needs the early part
    hipEvent_t e = pool_event(c, evi++);
    hipEventRecord(e, c->stream4);
    hipStreamWaitEvent(sm, e, 0);
  }
}
// A zeroed tile counter for a resident launch BEHIND the factorisation (the big levels of the inverse, K^-1 = W^T W): run_potrf clears the
// whole counter array once per factorisation, and these launches take the words of its top region in turn -- each used to clear its
// own word with a fill kernel on its stream first (6.5 us + a launch boundary, three of them on the tail of a cfg-2 evaluation).
static int* post_counter(hbo_ctx* c, int* counters, hipStream_t st) {
  constexpr int POST_POOL = 400;
  if (c->post_counter_next < POST_POOL) return counters + HBO_N_COUNTERS - 16 - c->post_counter_next++;
  int* p = counters + HBO_N_COUNTERS - 1;   // (pool exhausted: a word of its own, cleared on the spot)
  hipMemsetAsync(p, 0, sizeof(int), st);
  return p;
}
// The same level on the bf16 matrix cores (fp32, one matrix): both operands of each product are split exactly into three bf16
// planes (post3.hip) -- A: S21 = L21 W11 from row blocks of L and the transpose of W11, B: W21 = -W22 S21 from row blocks of
// W22 and the transpose of S21 -- then one launch per product over all groups of the level.
static bool trtri_level3(hbo_ctx* c, const TaskDesc* d_tasks, const TaskDesc& h, int max_nblk, int s, int grp_lo, int grp_hi,
                         bool do_a, bool do_b, hipStream_t st) {
  int ngrp = grp_hi - grp_lo;
  int vlast = std::min(s, max_nblk - ((grp_hi - 1) * 2 * s + s));
  if (vlast <= 0) { --ngrp; vlast = s; }
  if (ngrp <= 0) return true;
  const int nkb = 8 * s;
  // f16x2 form (run_potrf set it up: the caller knows max A_ii): L's blocks by the a-priori scale, W's and S21's by measured maxima
  // -- one word per level and operand; W11 / W22 are measured by a pass over what is about to be split, S21 by the product
  // that writes it.  The words only grow over the calls of a factorisation (other groups of the same level): still a bound.
  int li = 0;
  while ((1 << li) < s) ++li;
  unsigned int* const words = (c->h2_words && li < HBO_H2_LEVELS) ? c->h2_words + 3 * li : nullptr;
  const bool h2 = words != nullptr;
  const float sL = h2 ? post2h_scale_for(std::sqrt(c->chol_diag_bound)) : 1.f;
  const int64_t gstride = (int64_t)s * nkb * (h2 ? 2 : 3) * (HBO_TILE * 16);
  const size_t bytes = sizeof(unsigned short) * (size_t)gstride * ngrp;
  unsigned short* xp = static_cast<unsigned short*>(ws_get(c, WS_TRTRI3_X, bytes));
  unsigned short* yp = static_cast<unsigned short*>(ws_get(c, WS_TRTRI3_Y, bytes));
  if (!xp || !yp) return false;
  const int64_t ld = h.ld;
  const int64_t o0 = (int64_t)grp_lo * 2 * s * HBO_TILE, half = (int64_t)s * HBO_TILE;
  const float* L = static_cast<const float*>(h.A);
  const float* W = static_cast<const float*>(h.W);
  const float* S = static_cast<const float*>(h.S);
  Split3Block sb = {}; sb.ld = ld; sb.gstep = 2 * half * (ld + 1); sb.gstride = gstride; sb.nkb = nkb; sb.row_tiles = s;
  Syrk3Args g = {}; g.tasks = d_tasks; g.Xp = xp; g.Yp = yp; g.s = s; g.grp_lo = grp_lo; g.ngrp = ngrp; g.vlast = vlast;
  g.h2 = sb.h2 = h2 ? 1 : 0;
  const bool corun = st == c->stream4 && c->opt_trtri_free > 0 && c->trtri_counters;
  const int pblocks = 2 * (c->n_cus - c->opt_trtri_free);
  const int ntiles = ((ngrp - 1) * s + vlast) * s;
  auto launch = [&](int mode) {
    g.mode = mode; g.persistent = 0; g.work_counter = nullptr;
    g.yield_flag = (st == c->stream4) ? c->gemm_yield : nullptr;
    if (corun && ntiles > pblocks && c->trtri_counter_next < HBO_N_COUNTERS - HBO_N_BULK_COUNTERS - 512) { g.persistent = pblocks; g.work_counter = c->trtri_counters + c->trtri_counter_next++; }
    else if (!corun && c->opt_lauum_persist && ntiles >= 4 * c->n_cus) {   // behind the factorisation: as trtri_level's big levels
      int* counters = (int*)ws_get(c, WS_COUNTERS, sizeof(int) * HBO_N_COUNTERS);
      if (counters) {
        g.work_counter = post_counter(c, counters, st);
        g.persistent = 2 * c->n_cus;
      }
    }
    launch_syrk3(g, ntiles, 1, st);
  };
  ProfScope ps(c, "trtri_gemm", 2, st);
  if (do_a) {
    sb.in = L + (o0 + half) * ld + o0; sb.out = xp; sb.tri = 0; sb.last_rows = vlast; sb.last_krows = (int)half;
    sb.scale = sL; sb.scale_bits = nullptr; sb.max_out = nullptr;
    launch_split3_block(sb, ngrp, false, st);                     // rows of L21
    sb.in = W + o0 * ld + o0; sb.out = yp; sb.last_rows = s; sb.last_krows = (int)half;
    sb.max_out = h2 ? words + 0 : nullptr;
    launch_split3_block(sb, ngrp, true, st);                      // W11^T (the left half of every group is complete)
    g.sx = sL; g.sx_bits = nullptr; g.sy_bits = h2 ? words + 0 : nullptr; g.max_out = h2 ? words + 1 : nullptr;
    launch(1);
  }
  if (do_b) {
    sb.in = W + (o0 + half) * ld + (o0 + half); sb.out = xp; sb.tri = 1; sb.last_rows = vlast; sb.last_krows = vlast * HBO_TILE;
    sb.scale_bits = nullptr; sb.max_out = h2 ? words + 2 : nullptr;
    launch_split3_block(sb, ngrp, false, st);                     // rows of W22 (lower triangular)
    sb.in = S + (o0 + half) * ld + o0; sb.out = yp; sb.tri = 0; sb.last_rows = s; sb.last_krows = vlast * HBO_TILE;
    sb.scale_bits = h2 ? words + 1 : nullptr; sb.max_out = nullptr;
    launch_split3_block(sb, ngrp, true, st);                      // S21^T
    g.sx_bits = h2 ? words + 2 : nullptr; g.sy_bits = h2 ? words + 1 : nullptr; g.max_out = nullptr;
    launch(2);
  }
  return true;
}

// One level of the recursive inverse restricted to the groups [grp_lo, grp_hi) (a group = 2s blocks):
//   mode A: S21 = L21 W11,  mode B: W21 = -W22 S21.
static void trtri_level(hbo_ctx* c, int dtype, const TaskDesc* d_tasks, int ntasks, int max_nblk, int s,
                        int grp_lo, int grp_hi, bool do_a, bool do_b, hipStream_t st) {
  const TaskDesc& h_task0 = c->trtri_host_task;
  const int ngroups = grp_hi - grp_lo;
  if (ngroups <= 0) return;
  if (dtype == HBO_F32 && c->opt_trtri_bf16x3 && ntasks == 1 && h_task0.A && s >= c->opt_trtri3_min_s && max_nblk > small_limit(c, dtype) &&
      trtri_level3(c, d_tasks, h_task0, max_nblk, s, grp_lo, grp_hi, do_a, do_b, st))
    return;
  GemmArgs a = {}; a.tasks = d_tasks; a.p0 = s; a.grp_lo = grp_lo;
  // few or short tiles (small levels, small / batched matrices): 64x64 tiles -- a lone 128-tile runs
  // its K loop latency-bound, four 64-tiles expose 4x the parallelism for the same flops
  a.small_tiles = (max_nblk <= small_limit(c, dtype)) || ((int64_t)ngroups * s * s * ntasks < 600);
  a.yield_flag = (st == c->stream4) ? c->gemm_yield : nullptr;
  // products that co-run with the panel chain (single matrix, side stream): persistent, 2 workgroups on all but
  // `trtri_free` CUs, tiles from a counter -- see gemm_kernel
  const bool corun = st == c->stream4 && ntasks == 1 && c->opt_trtri_free > 0 && c->trtri_counters;
  // The dispatcher spreads a grid over the CUs breadth-first, so "free CUs" really means free room on every CU: a panel
  // kernel (potf2 78 KB, trsm 87 KB of LDS, 128 VGPRs) fits beside ONE 128-tile workgroup (72 KB) or TWO 64-tile
  // workgroups (2 x 40 KB), not beside more -- with 4 x (CUs - free) 64-tile workgroups every CU held three or four of
  // them and potf2 waited 330 us for the whole launch to end (rocprofv3 kernel trace, profiles/r02_potrf_chain.md)
  const int pblocks = 2 * (c->n_cus - c->opt_trtri_free);
  const int tmul = a.small_tiles ? 4 : 1;
  ProfScope ps(c, "trtri_gemm", 2, st);
  // Rows of the last group's lower half that exist in the largest task: workgroups beyond them would be dispatched
  // only to exit, which is not free (56 ns each: at the top level of a batch of 64 matrices of <= 19 blocks, 13 of
  // the 16 tile rows are empty and the launch took 4.3 ms instead of 0.8).
  const int vlast = std::min(s, max_nblk - ((grp_hi - 1) * 2 * s + s));
  if (vlast <= 0 && ngroups == 1) return;
  const int xa = (ngroups - 1) * s + std::max(vlast, 0);
  a.c_hi = grp_hi - 1; a.c_lo = std::max(vlast, 0);   // TRTRI_A: last group and its launched tile rows
  int post_slot = 0;
  auto persist = [&](int64_t tiles) {
    a.persistent = 0; a.work_counter = nullptr;
    if (corun && tiles > pblocks && c->trtri_counter_next < HBO_N_COUNTERS - HBO_N_BULK_COUNTERS - 512) { a.persistent = pblocks; a.work_counter = c->trtri_counters + c->trtri_counter_next++; }
    else if (!corun && ntasks == 1 && !a.small_tiles && c->opt_lauum_persist && tiles >= 4 * c->n_cus) {
      // behind the factorisation, one large matrix: the big levels as a resident grid drawing tiles from a counter, like K^-1 = W^T W
      // (run_lauum); the counters below the very last one are kept for this
      int* counters = (int*)ws_get(c, WS_COUNTERS, sizeof(int) * HBO_N_COUNTERS);
      if (counters && post_slot < 2) {
        a.work_counter = post_counter(c, counters, st); ++post_slot;
        a.persistent = 2 * c->n_cus;
      }
    }
  };
  if (do_a && xa > 0) { a.mode = GEMM_TRTRI_A; persist((int64_t)xa * s * tmul); a.tl = tl_slot("trtri_a", s); launch_gemm(dtype, a, dim3(xa, s, ntasks), st); }
  if (do_b) {
    a.mode = GEMM_TRTRI_B;
    const int vy = ngroups == 1 ? vlast : s;
    a.kt = vy;   // valid tile rows when there is a single group (blockIdx.y counts down from them)
    persist((int64_t)ngroups * s * vy * tmul);
    a.tl = tl_slot("trtri_b", s);
    launch_gemm(dtype, a, dim3(ngroups * s, vy, ntasks), st);
  }
}

// W = L^-1 by recursive doubling over the block tree: level s merges pairs of inverted s-block diagonal pieces,
//   S21 = L21 W11 (A),  W21 = -W22 S21 (B)   for every group g = blocks [2sg, 2sg + 2s).
// A of a group only needs the block columns below 2sg + s of L, B those below the group's end, so the tree can be
// walked while the factorisation is still running: trtri_advance(cfin) launches -- level by level, which is also the
// dependency order on one stream -- every piece that has become computable now that block columns [0, cfin) are
// final and was not launched before.  run_potrf calls it on a side stream after every fourth panel (the second half
// of the factorisation is bound by the serial panel chain and leaves most CUs idle); the last call, with
// cfin = max_nblk on the main stream, launches what is left (for a 64-block matrix: the B products on the right
// spine of the tree, 1.25 of the inverse's 3.7 ms).
void trtri_advance(hbo_ctx* c, int dtype, const TaskDesc* d_tasks, int ntasks, int max_nblk, int cfin,
                          hipStream_t st, TrtriProgress& pg, int max_s) {
  if (cfin > pg.diag) {
    ProfScope ps(c, "trtri_diag", 2, st);
    launch_trtri_diag(dtype, d_tasks, ntasks, pg.diag, cfin, st);
    pg.diag = cfin;
  }
  int li = 0;
  for (int s = 1; s < max_nblk && s <= max_s && li < 12; s *= 2, ++li) {
    // groups with a lower half: g*2s + s < max_nblk
    const int ngrp = (max_nblk - s + 2 * s - 1) / (2 * s);
    // A: left half final (cfin >= g*2s + s);  B: whole group final (cfin >= min(g*2s + 2s, max_nblk))
    int na = cfin >= s ? (cfin - s) / (2 * s) + 1 : 0;
    int nb = cfin >= max_nblk ? ngrp : cfin / (2 * s);
    na = std::min(na, ngrp); nb = std::min(nb, ngrp);
    if (na > pg.a[li]) { trtri_level(c, dtype, d_tasks, ntasks, max_nblk, s, pg.a[li], na, true, false, st); pg.a[li] = na; }
    if (nb > pg.b[li]) { trtri_level(c, dtype, d_tasks, ntasks, max_nblk, s, pg.b[li], nb, false, true, st); pg.b[li] = nb; }
  }
}
// ---- the one-sweep inverse -----------------------------------------------------------------------------------------------------
// With L = [[L11, 0], [L21, L22]] split at a row group R (q blocks): W = L^-1 = [[W11, 0], [-W22 L21 W11, W22]].  The product
// L21 W11 is not formed when R comes up -- it would be a skinny product with K up to N on the critical path of the sweep -- but
// accumulated: every finished group g' adds its term L[below, g'] W[g', :] to a running matrix T for ALL rows below it, a rank-q
// update of the same shape as the Cholesky's trailing update.  When R's block columns of L are final, T[R, :] is complete and
//   (a) W[R,R] = L[R,R]^-1        block-recursive inverse of the q x q diagonal group (trtri_advance capped at q / 2),
//   (b) W[R, <R] = -W[R,R] T[R, <R]                                                                  GEMM_SWEEP_B, K <= q
//   (c) T[>R, <=R] += L[>R, R] W[R, <=R]                                                              GEMM_SWEEP_T, K = q
//   (d) K^-1[<=R, <=R] += W[R, <=R]^T W[R, <=R]                                                       GEMM_SWEEP_C, K = q
// T and K^-1 share the S buffer: T lives below the front, K^-1 at and above it, and the rows R change roles between (b) and (d).
// Same flops as the recursive inverse + K^-1 = W^T W (N^3/3 each); what changes is WHEN they can run: everything but the last
// group's (a), (b), (d) sits beside the panel chain, and all of it is uniform K = 128 q work with thousands of tiles per launch.
bool use_sweep(const hbo_ctx* c, int dtype, int ntasks, int max_nblk) {
  if (!c->opt_sweep || !use_lookahead(c, ntasks, max_nblk) || max_nblk < 8) return false;
  if (c->opt_sweep >= 2) return true;
  // one matrix: at N = 8192 the rank-q updates of this form run at a lower rate than the long-K products of the recursive inverse
  // and of K^-1 = W^T W, and the phase has no idle machine to give them (13.4 against 11.5 ms in round 4's first measurement; with
  // the pipelined GEMM cores 11.14 against 10.71 at q = 16).  In between it wins -- fp64, ms per NLL + gradient, recursive / sweep:
  //   N = 2048 1.193 / 1.193, 2560 1.580 / 1.417 (q = 4), 3072 1.944 / 1.740 (4), 3584 2.261 / 2.167 (4), 4096 2.780 / 2.671 (8),
  //   5120 4.404 / 4.179 (8), 6144 6.053 / 6.010 (8)        (profiles/r04_gemm_pipeline.md)
  if (ntasks == 1) return dtype == HBO_F64 && max_nblk > 16 && max_nblk <= 48;
  // fp32 beyond the small sizes: the block-recursive products and K^-1 run on the bf16 matrix cores (post3.hip), which this form does not
  if (dtype == HBO_F32 && ntasks == 1 && max_nblk > small_limit(c, dtype) && (c->opt_trtri_bf16x3 || c->opt_lauum_bf16x3)) return false;
  return true;
}
int sweep_group(int ntasks, int max_nblk) { return (ntasks == 1 && max_nblk > 28) ? 8 : 4; }
static hipEvent_t sweep_event(hbo_ctx* c, SweepState& sw) {
  const size_t i = (size_t)sw.nev++;
  while (c->ev_pool_sweep.size() <= i) {
    hipEvent_t ev;
    hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    c->ev_pool_sweep.push_back(ev);
  }
  return c->ev_pool_sweep[i];
}
void sweep_advance(hbo_ctx* c, int dtype, const TaskDesc* d_tasks, int ntasks, int max_nblk, int cfin, hipStream_t st, SweepState& sw,
                   hipEvent_t w_done) {
  const int q = sw.qs;
  // 64-tiles for small / batched matrices as everywhere else -- except where a launch has enough 128-tiles to fill the machine
  // several times over: all tiles of a sweep launch have the same K, so the larger tile's better MFMA rate is not lost in a tail
  const bool small_shape = max_nblk <= small_limit(c, dtype);
  bool small = small_shape;
  int U = small ? 2 : 1;
  auto pick = [&](int64_t tiles128) { small = small_shape && tiles128 * ntasks < c->opt_sweep_big; U = small ? 2 : 1; };
  // beside the panel chain (side stream, counters of this factorisation at hand): persistent and slot-limited, tiles (x tasks) from
  // a counter, polling the yield table when the chain's kernels keep one -- see trtri_level
  const bool corun = st == c->stream4 && c->trtri_counters && c->opt_trtri_free > 0 && (ntasks == 1 || batch_bg(c, ntasks) >= 1);
  // (the sweep's launches leave fewer CUs free than the recursive inverse's big products: N = 5120 3.785 -> 3.70 ms at 16-32 instead of 48)
  const int pblocks = 2 * (c->n_cus - std::min(c->opt_trtri_free, c->opt_sweep_free));
  auto place = [&](GemmArgs& a, int64_t tiles) {
    a.persistent = 0; a.work_counter = nullptr;
    a.yield_flag = (st == c->stream4) ? c->gemm_yield : nullptr;
    if (corun && tiles * ntasks > pblocks && c->trtri_counter_next < HBO_N_COUNTERS - HBO_N_BULK_COUNTERS - 512) {
      a.persistent = pblocks; a.work_counter = c->trtri_counters + c->trtri_counter_next++;
    }
  };
  // (d) has no successor but the next group's (d) and the final consumers of K^-1: on a stream of its own it runs beside the
  // (a) -> (b) -> (c) chain of the following groups instead of holding it up.  Measured (ms, same stream / own stream): one matrix on
  // 128-tiles N = 5120 4.26 / 4.15, 6144 6.07 / 5.71, 7168 8.39 / 7.96, 8192 11.44 / 10.97; on 64-tiles N = 2560 1.43 / 1.48,
  // 4096 2.65 / 2.79; batches 14.11 / 14.48 (64 tasks), 2.57 / 2.74 (8): where the launches are small the extra hop costs more
  hipStream_t sd = (c->opt_sweep_side && c->stream3 && ntasks == 1 && max_nblk > small_limit(c, dtype)) ? c->stream3 : st;
  while (sw.done < max_nblk) {
    const int b0 = sw.done, b1 = std::min(b0 + q, max_nblk);
    if (b1 > cfin) break;
    if (sw.ev_c && sw.st_c != st) hipStreamWaitEvent(st, sw.ev_c, 0);   // T[R, <R] is complete (and every earlier launch of that stream)
    trtri_advance(c, dtype, d_tasks, ntasks, max_nblk, b1, st, sw.pg, q / 2);                       // (a)
    GemmArgs a = {}; a.tasks = d_tasks; a.c_lo = b0; a.c_hi = b1;
    if (b0 > 0) {                                                                                    // (b)
      ProfScope ps(c, "sweep_b", 2, st);
      pick((int64_t)b0 * (b1 - b0)); a.small_tiles = small;
      a.mode = GEMM_SWEEP_B; place(a, (int64_t)b0 * (b1 - b0) * U * U); a.tl = tl_slot("sweep_b", b1);
      launch_gemm(dtype, a, dim3(b0, b1 - b0, ntasks), st);
    }
    if (b1 == max_nblk && w_done) hipEventRecord(w_done, st);
    if (sd != st) { hipEvent_t e = sweep_event(c, sw); hipEventRecord(e, st); hipStreamWaitEvent(sd, e, 0); }   // W[R, <=R] is final
    if (b1 < max_nblk) {                                                                             // (c)
      ProfScope ps(c, "sweep_t", 2, st);
      pick((int64_t)(max_nblk - b1) * b1); a.small_tiles = small;
      a.mode = GEMM_SWEEP_T; place(a, (int64_t)(max_nblk - b1) * b1 * U * U); a.tl = tl_slot("sweep_t", b1);
      launch_gemm(dtype, a, dim3(max_nblk - b1, b1, ntasks), st);
      sw.ev_c = sweep_event(c, sw); hipEventRecord(sw.ev_c, st); sw.st_c = st;
    }
    {                                                                                                // (d)
      ProfScope ps(c, "sweep_c", 2, sd);
      pick((int64_t)b1 * (b1 + 1) / 2); a.small_tiles = small;
      a.mode = GEMM_SWEEP_C; place(a, small ? 2 * (int64_t)b1 * (b1 + 1) : (int64_t)b1 * (b1 + 1) / 2);
      if (sw.ev_d && sw.st_d != sd) hipStreamWaitEvent(sd, sw.ev_d, 0);   // the previous group's update of the same K^-1 tiles
      a.tl = tl_slot("sweep_c", b1);
      launch_gemm(dtype, a, dim3(b1, 1, ntasks), sd);
      if (b1 < max_nblk) { sw.ev_d = sweep_event(c, sw); hipEventRecord(sw.ev_d, sd); sw.st_d = sd; }
    }
    sw.done = b1;
    if (b1 == max_nblk && sd != st) { hipEvent_t e = sweep_event(c, sw); hipEventRecord(e, sd); hipStreamWaitEvent(st, e, 0); }   // K^-1 is complete
  }
}
void run_trtri(hbo_ctx* c, int dtype, const TaskDesc* d_tasks, int ntasks, int max_nblk, TrtriProgress* pg) {
  TrtriProgress fresh;
  trtri_advance(c, dtype, d_tasks, ntasks, max_nblk, max_nblk, c->stream, pg ? *pg : fresh);
}
// K^-1 = W^T W on the lower tiles.  (A two-launch form that started the W11^T W11 part beside the tail of the inverse was
// built and measured neutral in round 2 -- profiles/r02_potrf_chain.md -- and is gone.)
void run_lauum(hbo_ctx* c, int dtype, const TaskDesc* d_tasks, int ntasks, int max_nblk, hipStream_t st) {
  ProfScope ps(c, "lauum", 2, st ? st : c->stream);
  // fp32, one matrix beyond the small sizes: on the bf16 matrix cores from ONE exact three-way split of W^T (post3.hip,
  // syrk3_kernel mode 3) -- 6 bytes per element of the lower triangle of W as workspace
  const TaskDesc& h = c->trtri_host_task;
  if (dtype == HBO_F32 && c->opt_lauum_bf16x3 && ntasks == 1 && h.W && h.nblk == max_nblk && max_nblk > small_limit(c, dtype)) {
    const int nkb = 8 * max_nblk;
    const size_t bytes = sizeof(unsigned short) * (size_t)max_nblk * nkb * 3 * (HBO_TILE * 16);
    unsigned short* xp = static_cast<unsigned short*>(ws_get(c, WS_LAUUM3, bytes));
    if (xp) {
      hipStream_t s = st ? st : c->stream;
      const int n = max_nblk * HBO_TILE;
      Syrk3Args g = {}; g.tasks = d_tasks; g.Xp = xp; g.nkb = nkb; g.mode = 3;
      if (c->h2_words) {   // the f16x2 form: max |W| by one pass, W^T as two fp16 planes, three MFMAs per product
        unsigned int* word = c->h2_words + HBO_H2_WORDS - 1;
        launch_split2h_transpose_measured(static_cast<const float*>(h.W), h.ld, n, n, xp, nkb, word, s, 1);
        g.h2 = 1; g.sx_bits = g.sy_bits = word;
      } else launch_split3_transpose(static_cast<const float*>(h.W), h.ld, n, n, xp, nkb, s, 1);
      const int nt = max_nblk * (max_nblk + 1) / 2;
      int* counters = c->opt_lauum_persist && nt > 4 * c->n_cus ? (int*)ws_get(c, WS_COUNTERS, sizeof(int) * HBO_N_COUNTERS) : nullptr;
      if (counters) {   // a resident grid drawing the tiles from a counter, as the fp64 form below
        g.work_counter = post_counter(c, counters, s);
        g.persistent = 2 * (c->n_cus - (c->opt_lauum_persist > 1 ? c->opt_lauum_persist : 16));
      }
      launch_syrk3(g, nt, 1, s);
      return;
    }
  }
  GemmArgs a = {}; a.tasks = d_tasks; a.mode = GEMM_LAUUM;
  a.small_tiles = max_nblk <= small_limit(c, dtype);
  hipStream_t s = st ? st : c->stream;
  if (ntasks == 1 && !a.small_tiles && c->opt_lauum_persist && max_nblk * (max_nblk + 1) / 2 > 4 * c->n_cus) {
    // one large matrix: a resident grid of two workgroups per CU draws the tiles from a counter (gemm.hip: launch_gemm_t)
    int* counters = (int*)ws_get(c, WS_COUNTERS, sizeof(int) * HBO_N_COUNTERS);
    if (counters) {
      a.work_counter = post_counter(c, counters, s);
      // (16 CUs stay free for alpha = W^T z and d nll / d mu, which run beside this launch on the panel stream: isolated the
      //  launch takes the same time with 480 as with 512 workgroups)
      a.persistent = 2 * (c->n_cus - (c->opt_lauum_persist > 1 ? c->opt_lauum_persist : 16));
    }
  }
  a.tl = tl_slot("lauum", 0);
  launch_gemm(dtype, a, dim3(max_nblk, max_nblk, ntasks), s);
}
