// Context object of libhbo and the error plumbing shared by its host translation units (api.hip: the C ABI and the
// orchestration of the kernels; comm.hip: the RCCL binding).
#pragma once
#include "hbo_internal.h"

#include <stdio.h>

#include <map>
#include <string>
#include <utility>
#include <vector>

extern thread_local std::string hbo_g_err;   // last error of ctx-less calls (hbo_last_error(NULL))

struct ProfEntry { const char* name; hipEvent_t e0, e1; };

struct hbo_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  hipStream_t stream2 = nullptr;   // panel stream of the look-ahead Cholesky (high priority)
  hipStream_t stream3 = nullptr;   // the later block columns of F1 (run_potrf: split_f1), beside the next diagonal block's factorisation and solve
  hipStream_t stream4 = nullptr;   // early part of trtri, overlapped with the tail of potrf
  int opt_cu_yield = 2;      // background GEMM workgroups pause while a panel-chain workgroup runs on their CU (single matrix,
                             // look-ahead): 1 = potf2 only, 2 = trsm and the chain's column updates too
  int* d_yield = nullptr;    // per-CU table (cu_token() -> panel-chain workgroups running there)
  int* gemm_yield = nullptr; // run_potrf -> trtri_level: GemmArgs::yield_flag of the launches that co-run with the panel chain
  int opt_small_nblk = -1;   // matrices up to this many 128-blocks use 64x64 GEMM tiles in trtri / lauum / the sweep (-1: auto, sched.hip: small_limit)
  int opt_persist_free = -1; // bulk trailing update runs as 2*(CUs - this) persistent workgroups (-1: auto, see run_potrf)
  int opt_post_chunk = 8192;   // posterior / acquisition: candidates per pass (the cross-Gram workspace is npad x this, whatever M)
  int opt_trtri_bf16x3 = 1;    // fp32, one matrix: the products of the block-recursive inverse on the bf16 cores from level trtri3_min_s on
  int opt_trtri3_min_s = 8;
  int opt_lauum_bf16x3 = 1;    // fp32, one matrix: K^-1 = W^T W on the bf16 cores too
  TaskDesc trtri_host_task = {};   // host copy of the single task's descriptor (pointers, ld) for those launches; valid when .A != null
  int opt_syrk3_col = 0;       // 1: the left-looking column updates inside a group on the bf16 cores too (needs a split per panel)
  int opt_syrk3_sep = 0;       // debug: 1 = the panels are split by a kernel of their own instead of inside the panel solve
  int opt_syrk3_free = 32;     // ... CUs the bulk update of that form leaves with a single workgroup (room for the panel kernels)
  int opt_syrk_bf16x3 = 1;     // fp32 factorisations: trailing updates on the bf16 matrix cores (exact three-way split of the panels, post3.hip)
  int opt_chol_f16x2 = 1;      // fp32 factorisations of the stationary covariances: trailing updates, inverse levels and K^-1 = W^T W on two-way fp16 splits
                               // (three MFMAs per product instead of bf16x3's six; needs chol_diag_bound, i.e. a caller that knows max_i A_ii)
  unsigned int* h2_words = nullptr;   // run_potrf -> trtri_level3 / run_lauum: measured operand maxima of the f16x2 products (null: bf16x3)
  double chol_diag_bound = 0;  // set by the objective / factor paths around run_potrf: max_i A_ii (signal variance + noise + jitter); 0: unknown
  int opt_post_f16x2 = 1;      // fp32 posterior product of the stationary covariances: two-way fp16 split, three MFMAs per product (post2h.hip) instead of bf16x3's six
  int opt_post_bf16x3 = 1;     // fp32 posterior product on the bf16 matrix cores (three-way exact split of both operands, post3.hip); 0: fp32 MFMA
  int opt_trtri_at = 0;        // single matrix: panel count (in 64ths of the block count) after which the inverse starts beside the chain (0: 5/8)
  int opt_sweep = 1;           // one-sweep inverse (sched.hip:sweep_advance): 0 never, 1 where measured faster (use_sweep), 2 wherever look-ahead is on
  int opt_sweep_big = 4000;    // a sweep launch of a small / batched shape with at least this many 128-tiles (x tasks) runs on 128-tiles
  int opt_sweep_side = 1;      // one matrix on 128-tiles: the sweep's K^-1 updates (d) on a stream of their own beside its (a) (b) (c) chain
  int opt_sweep_free = 24;     // CUs the sweep's persistent launches beside the chain leave with one workgroup (at most trtri_free)
  int opt_sweep_qs = 0;        // its row-group size in 128-blocks (power of two; 0: auto)
  int opt_batch_bg = -1;       // batches: the sweep's launches beside the panel chain are 0 plain grids, 1 persistent and slot-limited (-1: auto = 1 up to 8 tasks),
                               // (tiles x tasks from one counter), 2 also yielding to the chain's kernels through the per-CU table
  int opt_poison = 0;          // tests: every evaluation first fills what it is about to recompute with NaN (gram.hip: poison_kernel)
  int opt_f2_split = 0;        // hbo_tune("f2_split"): the bulk update F2 in two launches -- the columns the NEXT F1 accumulates into first, with the
                               // event behind them -- so that the panel chain runs up to one bulk launch ahead (1: one matrix, 2: batches too)
  int opt_split_f1 = 1;        // panel chain: F1 updates only the next block column on the panel stream, the group's later columns on a third stream:
                               // 0 never, 1 for batches (where it was measured faster), 2 always
  int opt_lauum_persist = 1;   // one large matrix: K^-1 = W^T W as a resident grid drawing its tiles from a counter (0: plain grid; 1: two workgroups
                               // on all but 16 CUs; n > 1: on all but n CUs)
  int opt_trtri_free = 48;   // CUs the inverse products that co-run with the panel chain leave free (0: one tile per workgroup)
  int post_counter_next = 0;   // sched.hip: post_counter
  int* trtri_counters = nullptr; int trtri_counter_next = 0;   // run_potrf -> trtri_level: tile counters of those launches
  int n_cus = 256;
  std::vector<hipEvent_t> ev_pool;
  std::vector<hipEvent_t> ev_pool_sweep;   // sweep_advance's own events (its calls interleave with run_potrf's use of ev_pool)
  hipEvent_t ev_timed[3] = {nullptr, nullptr, nullptr};   // hbo_objective_sharded: start / local shard done / all-reduce done
  std::map<int, std::pair<void*, size_t>> ws;   // grow-only device scratch buffers by slot (no per-call hipMalloc/hipFree)
  // Device buffers of freed datasets / caches, kept for the next one of the same shape (dev_alloc / dev_free in api.hip):
  // GP.train()'s Adam loop re-creates its sub-sampled batch every step (gp.py:101-111) and 64 tasks x 8 buffers of
  // hipMalloc + hipFree cost 40 of its 50 ms.  Key: (class, bytes); class 1 / 2 = an fp32 / fp64 inverse factor W, whose
  // "zeros above the diagonal" survive a reuse in the same role only.
  std::map<std::pair<int, size_t>, std::vector<void*>> pool_free;
  std::map<void*, std::pair<int, size_t>> pool_live;
  size_t pool_bytes = 0;                       // bytes parked in pool_free
  size_t pool_cap = (size_t)48 << 30;          // lowered to a quarter of the device memory at context creation
  int opt_lookahead = 1;
  int opt_overlap_trtri = 1;
  std::string err;
  ModelDev* h_model = nullptr;      // pinned: uploaded without a staging copy or a synchronisation
  void* hp_stage = nullptr; size_t hp_stage_bytes = 0;   // pinned staging (descriptors up, results down)
  hipEvent_t ev_upload = nullptr;   // last host->device copy out of the pinned buffers
  ModelDev* d_model = nullptr;
  void* d_mlp_w[HBO_MAX_MLP_LAYERS] = {nullptr};
  void* d_mlp_b[HBO_MAX_MLP_LAYERS] = {nullptr};
  size_t mlp_w_bytes[HBO_MAX_MLP_LAYERS] = {0};
  size_t mlp_b_bytes[HBO_MAX_MLP_LAYERS] = {0};
  int opt_group_inner = -1;  // two-level panel groups: column updates on the chain inside inner groups of this many panels (0: one level; -1: auto, see run_potrf)
  int opt_group = 0;         // 128-wide panels per trailing update (K = 128*group); 0: auto, see run_potrf
  int prof_level = 0;
  std::vector<ProfEntry> prof_pending;
  std::vector<hipEvent_t> prof_events; size_t prof_next = 0;   // event pool of the timing scopes
  std::vector<std::string> prof_names;
  std::vector<double> prof_ms;
  std::vector<int> prof_count;
  // RCCL
  void* rccl_lib = nullptr;
  void* comm = nullptr;
  int lds_per_block = 160 * 1024;   // hipDeviceProp_t::sharedMemPerBlock of the context's device (hbo_ctx_create)
  int opt_small_fused = 1;     // hbo_tune("small_fused"): batches whose tasks all have n <= 128 take the single-workgroup evaluation (small.hip)
  int opt_post_serial = 0;     // hbo_tune("post_serial"): the streamed posterior's producer side (cross Gram) on the SAME stream as its products: isolated stage times
  int opt_fault_shard = 0;     // hbo_tune("fault_shard"): ONE-SHOT fault injection for the tests of the sharded objective's failure paths
  bool comm_aborted = false;   // set by comm_abort: sharded calls fail with HBO_ERR_COMM until hbo_comm_init builds a new communicator
  double* d_comm_buf = nullptr;
  int comm_buf_count = 0;
};

#define HIPCHK(ctx, call)                                                                     \
  do {                                                                                        \
    hipError_t e__ = (call);                                                                  \
    if (e__ != hipSuccess) {                                                                  \
      char buf__[512];                                                                        \
      snprintf(buf__, sizeof buf__, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e__),   \
               __FILE__, __LINE__);                                                           \
      if (ctx) (ctx)->err = buf__; else hbo_g_err = buf__;                                        \
      return HBO_ERR_HIP;                                                                     \
    }                                                                                         \
  } while (0)

static inline int fail(hbo_ctx* ctx, int code, const std::string& msg) {
  if (ctx) ctx->err = msg; else hbo_g_err = msg;
  return code;
}
