// Device-side counter protocol shared by the resident tile-task kernel (dag.hip) and the panel-chain kernels (chol.hip).
// Visibility follows the agent-scope release / acquire rule for gfx950 (per-XCD L2s, per-CU L1s): a producer drains its
// stores, ONE lane runs the release fence (L2 write-back) and bumps the counter with a relaxed agent-scope atomic; a
// consumer polls that word relaxed and runs ONE acquire fence (L1 invalidate) after the match, then reads with plain loads.
// Every wait is bounded by a wall-clock budget; the first waiter that runs out of it raises DAG_ABORT and everybody leaves.
#pragma once
#include "dag.h"

__device__ __forceinline__ int dag_load(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void dag_bump(int* p, int inc) { __hip_atomic_fetch_add(p, inc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// A whole wave in lockstep: wait until ctr[idx] >= thr.  false: aborted (by this waiter's timeout or by somebody else's).
// (Not "one lane polls while the others wait": a spin loop under a lane mask inside a loop that also holds workgroup barriers
// is structurised into a deadlock, see dag.hip:dag_next_task.)
__device__ __forceinline__ bool dag_wait_ge(int* ctr, int idx, int thr, long long timeout_ticks) {
  if (dag_load(ctr + idx) >= thr) return true;
  const unsigned long long t0 = wall_clock64();
  for (int spin = 0;; ++spin) {
    __builtin_amdgcn_s_sleep(4);
    if (dag_load(ctr + idx) >= thr) return true;
    if (dag_load(ctr + DAG_ABORT)) return false;
    if ((spin & 255) == 255 && (long long)(wall_clock64() - t0) > timeout_ticks) {
      if ((threadIdx.x & 63) == 0) __hip_atomic_store(ctr + DAG_ABORT, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return false;
    }
  }
}
// every wave of the workgroup on its own: polls, then acquires (no LDS, no workgroup barrier)
__device__ __forceinline__ bool dag_wave_wait(int* ctr, int idx, int thr, long long timeout_ticks) {
  const bool ok = dag_wait_ge(ctr, idx, thr, timeout_ticks);
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  return ok;
}
// the whole workgroup: every wave drains its stores, then wave 0 releases and its lane 0 bumps
__device__ __forceinline__ void dag_wg_publish(int* ctr, int idx, int inc) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x < 64) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (threadIdx.x == 0) dag_bump(ctr + idx, inc);
  }
}

#ifdef HBO_DAG_DEBUG
__device__ __forceinline__ void dag_stamp_min(unsigned long long* st, int p, int k) {
  if (st) __hip_atomic_fetch_min(st + p * DAG_STAMPS_PER_PANEL + k, (unsigned long long)wall_clock64(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void dag_stamp_max(unsigned long long* st, int p, int k) {
  if (st) __hip_atomic_fetch_max(st + p * DAG_STAMPS_PER_PANEL + k, (unsigned long long)wall_clock64(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
#else
#define dag_stamp_min(st, p, k) do {} while (0)
#define dag_stamp_max(st, p, k) do {} while (0)
#endif
