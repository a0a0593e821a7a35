// Launch schedules of the blocked factorisation, the block-recursive inverse and K^-1 = W^T W (sched.hip).
#pragma once
#include "runtime.h"

// progress of the block-recursive inverse (see trtri_advance)
struct TrtriProgress { int diag = 0; int a[12] = {0}; int b[12] = {0}; };

hipEvent_t pool_event(hbo_ctx* c, size_t i);
// h_nblk (optional): block count per task, host copy -- a batch needs it for the resident tile-task schedule
void run_potrf(hbo_ctx* c, int dtype, const TaskDesc* d_tasks, int ntasks, int max_nblk, int* d_info, TrtriProgress* early = nullptr,
               const int* h_nblk = nullptr);
// dag.hip: the same factorisation as one resident tile-task kernel beside the panel kernels; false: not applicable here
bool run_potrf_dag(hbo_ctx* c, int dtype, const TaskDesc* d_tasks, int ntasks, int max_nblk, const int* h_nblk, int* d_info,
                   TrtriProgress* early);
// after the stream synchronisation: did the last resident run hit its wall-clock bound?  (the caller then repeats the
// evaluation on the launch schedule; the context stays there)
bool dag_aborted(hbo_ctx* c);
void trtri_advance(hbo_ctx* c, int dtype, const TaskDesc* d_tasks, int ntasks, int max_nblk, int cfin, hipStream_t st, TrtriProgress& pg);
void run_trtri(hbo_ctx* c, int dtype, const TaskDesc* d_tasks, int ntasks, int max_nblk, TrtriProgress* pg = nullptr);
void run_lauum(hbo_ctx* c, int dtype, const TaskDesc* d_tasks, int ntasks, int max_nblk, hipStream_t st = nullptr);
