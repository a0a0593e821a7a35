// Launch schedules of the blocked factorisation, the block-recursive inverse and K^-1 = W^T W (sched.hip).
#pragma once
#include "runtime.h"

// progress of the block-recursive inverse (see trtri_advance)
struct TrtriProgress { int diag = 0; int a[12] = {0}; int b[12] = {0}; };

// The one-sweep inverse (sweep_advance): W = L^-1 and K^-1 = W^T W built row group by row group BEHIND the panel chain, so that
// what is left when the factorisation ends is one group's worth of O(N^2 q) work instead of the O(N^3) tail of the block-recursive
// inverse and of K^-1 = W^T W (which needs every row of W).
struct SweepState {
  int qs = 4; int done = 0; TrtriProgress pg;
  int nev = 0;                                   // events of ev_pool_sweep used so far
  // behind the last processed group's (c) and (d), and the streams they ran on: a later call on ANOTHER stream (the tail behind the
  // factorisation, on the main stream) waits for (c) before its (a) / (b) and for (d) only before its own (d)
  hipEvent_t ev_c = nullptr, ev_d = nullptr;
  hipStream_t st_c = nullptr, st_d = nullptr;
};
int sweep_group(int ntasks, int max_nblk);   // row-group size of the one-sweep inverse where use_sweep() says yes

hipEvent_t pool_event(hbo_ctx* c, size_t i);
// Look-ahead (panel chain, bulk update and inverse on separate streams) pays once there is something to overlap; below that the
// events and cross-stream waits cost more than they hide.  Measured (NLL+grad, look-ahead on / off, ms): one matrix of 2 / 8 / 16 / 20 /
// 24 blocks: 0.282 / 0.253, 0.714 / 0.682, 1.362 / 1.328, 1.724 / 1.741, 2.154 / 2.204; 64 tasks of 2 / 4 / 6 / 12 blocks: 0.344 / 0.298,
// 0.790 / 0.761, 1.569 / 1.598, 6.89 / 7.05; 8 tasks of 4 / 8 / 12 / 16-19 blocks: 0.457 / 0.415, 0.847 / 0.827, 1.508 / 1.591, 3.34 / 3.61.
// Option lookahead: 0 = never, 1 = this rule (default), 2 = whenever there is more than one block.
static inline bool use_lookahead(const hbo_ctx* c, int ntasks, int max_nblk) {
  if (!c->opt_lookahead || max_nblk <= 1) return false;
  if (c->opt_lookahead >= 2) return true;
  return max_nblk > 4 && (max_nblk >= 18 || (int64_t)ntasks * max_nblk >= 80);
}
void run_potrf(hbo_ctx* c, int dtype, const TaskDesc* d_tasks, int ntasks, int max_nblk, int* d_info, TrtriProgress* early = nullptr,
               SweepState* sweep = nullptr);
// does the objective pipeline of this shape take the one-sweep inverse?  (sched.hip)
bool use_sweep(const hbo_ctx* c, int dtype, int ntasks, int max_nblk);
// every row group whose block columns [.., cfin) are final and that was not swept before; `w_done` (optional): recorded on `st`
// as soon as W is complete (behind the last group's rows), so that alpha = W^T z can start beside the last K^-1 update
void sweep_advance(hbo_ctx* c, int dtype, const TaskDesc* d_tasks, int ntasks, int max_nblk, int cfin, hipStream_t st, SweepState& sw,
                   hipEvent_t w_done = nullptr);
void trtri_advance(hbo_ctx* c, int dtype, const TaskDesc* d_tasks, int ntasks, int max_nblk, int cfin, hipStream_t st, TrtriProgress& pg,
                   int max_s = 1 << 30);
void run_trtri(hbo_ctx* c, int dtype, const TaskDesc* d_tasks, int ntasks, int max_nblk, TrtriProgress* pg = nullptr);
void run_lauum(hbo_ctx* c, int dtype, const TaskDesc* d_tasks, int ntasks, int max_nblk, hipStream_t st = nullptr);
