// Persistent panel-chain kernel of the blocked Cholesky (single task, look-ahead mode).
//
// The factorisation's critical path -- per 128-column panel: left-looking update of the block column, potf2 of the
// diagonal block, trsm of the rows below -- used to be three kernel launches per panel that had to find free workgroup
// slots between the workgroups of the trailing-update GEMMs (profiles/r01_potrf_chain.md: slot waiting, 3x slower
// kernels beside the bulk update, a 0.8 ms stall when the early inverse starts).  Here `nwg` workgroups are launched
// once and stay resident for the whole factorisation; each requests more than half of a CU's LDS, so a workgroup owns
// its CU (no trailing-update workgroup fits beside it) and the chain never waits for placement.  Steps inside a panel
// are separated by a grid barrier (atomic counter, agent-scope release/acquire), not by kernel boundaries.
//
// The trailing updates stay host-launched GEMMs on the other CUs and are tied to the chain with signal memory:
//   s_panels : the chain adds 1 when a group of q panels is final      -> hipStreamWaitValue64 releases F2(g)
//   s_bulk   : hipStreamWriteValue64 after F2(g) completes             -> the chain spins on it before it touches
//              block columns that F2 wrote (group g + 2 onwards).
// Block column p is brought up to date left-looking inside the chain: K = [start of the previous group, p), i.e. the
// previous group's panels (F2 skips the next group's columns) plus the earlier panels of its own group.
#include "hbo_internal.h"
#include <limits.h>
#define HBO_DEVICE_ONLY
namespace hbo_gemm {
#include "gemm.hip"
}
namespace hbo_chol {
#include "chol.hip"
}

namespace {

struct ChainSync { unsigned count; unsigned gen; unsigned potf2_done; unsigned timed_out; };

// Every spin is bounded (s_memrealtime, 100 MHz): a protocol error must end in wrong numbers -- which the caller
// detects through `timed_out` -- never in a kernel that holds the GPU.
constexpr unsigned long long SPIN_LIMIT = 300000000ull;   // 3 s
#define HBO_SPIN_WHILE(cond)                                                         \
  do {                                                                               \
    const unsigned long long t_spin0 = wall_clock64();                               \
    while (cond) {                                                                   \
      __builtin_amdgcn_s_sleep(1);                                                   \
      if (wall_clock64() - t_spin0 > SPIN_LIMIT) { atomicExch(&sy->timed_out, 1u); break; } \
    }                                                                                \
  } while (0)

__device__ __forceinline__ void grid_barrier(ChainSync* sy, unsigned nwg, unsigned& gen_local) {
  __threadfence();                       // release this workgroup's global writes (agent scope: L2 write-back)
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned target = gen_local + 1;
    if (atomicAdd(&sy->count, 1u) == nwg - 1) {
      atomicExch(&sy->count, 0u);
      __threadfence();
      atomicExch(&sy->gen, target);
    } else {
      HBO_SPIN_WHILE(__hip_atomic_load(&sy->gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != target &&
                     !__hip_atomic_load(&sy->timed_out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    }
  }
  ++gen_local;
  __syncthreads();
  __threadfence();                       // acquire: later plain loads see the other workgroups' writes
}

#ifdef HBO_CHAIN_TIMING
__device__ unsigned long long hbo_dbg_chain[6 * 256];   // per panel (workgroup 0): start, waited, updated, factored, solved, end
#define CSTAMP(k) do { if (wg == 0 && threadIdx.x == 0 && p < 256) hbo_dbg_chain[6 * p + (k)] = wall_clock64(); } while (0)
#else
#define CSTAMP(k) do {} while (0)
#endif
template <typename T>
__global__ __launch_bounds__(256) void chain_kernel(ChainArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int BKE = 128 / sizeof(T);
  const TaskDesc& t = a.tasks[0];
  const int nblk = t.nblk;
  const int nwg = gridDim.x, wg = blockIdx.x;
  const int64_t ld = t.ld;
  ChainSync* sy = reinterpret_cast<ChainSync*>(a.sync);
  unsigned gen = 0;
  T* Am = static_cast<T*>(t.A);
  const int nrt64 = (nblk + 1) * 2;      // 64-row tiles incl. the augmented tile-row

  for (int p = 0; p < nblk; ++p) {
    const int grp = p / a.q, g0 = grp * a.q;
    CSTAMP(0);
    // block columns of group >= 2 carry the bulk update of group - 2: wait for it before touching them
    if (p == g0 && grp >= 2) {
      if (threadIdx.x == 0) {
        const unsigned long long need = a.bulk_base + (unsigned long long)(grp - 1);
        HBO_SPIN_WHILE(__hip_atomic_load(a.s_bulk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < need &&
                       !__hip_atomic_load(&sy->timed_out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
      }
      __syncthreads();
      __threadfence();
    }
    CSTAMP(1);
    // ---- left-looking update of block column p: K = [max(0, g0 - q), p) ---------------------------------------
    const int kbeg = g0 - a.q > 0 ? g0 - a.q : 0;
    if (p > kbeg) {
      hbo_gemm::TileJob<T> job;
      job.lda = job.ldb = job.ldc = ld;
      job.colsq = nullptr;
      job.ksteps = (p - kbeg) * HBO_TILE / BKE;
      job.alpha = (T)-1; job.beta = 1;
      const int c0 = 2 * p;
      int tix = 0;
      for (int r = c0; r < nrt64; ++r)
        for (int ch = 0; ch < 2; ++ch) {
          const int c = c0 + ch;
          if (r < c) continue;
          if (tix++ % nwg != wg) continue;
          job.A = Am + (int64_t)r * 64 * ld + (int64_t)kbeg * HBO_TILE;
          job.B = Am + (int64_t)c * 64 * ld + (int64_t)kbeg * HBO_TILE;
          job.C = Am + (int64_t)r * 64 * ld + (int64_t)c * 64;
          hbo_gemm::gemm_tile<T, true, true, 64>(job, smem);
          __syncthreads();
        }
      grid_barrier(sy, nwg, gen);
    }
    CSTAMP(2);
    // ---- potf2 of the diagonal block (workgroup 0), the others wait for its flag ------------------------------
    if (wg == 0) {
      hbo_chol::potf2_body<T>(t, p, a.info, smem);
      __threadfence();
      __syncthreads();
      if (threadIdx.x == 0) atomicExch(&sy->potf2_done, (unsigned)(p + 1));
    } else {
      if (threadIdx.x == 0)
        HBO_SPIN_WHILE(__hip_atomic_load(&sy->potf2_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(p + 1) &&
                       !__hip_atomic_load(&sy->timed_out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
      __syncthreads();
      __threadfence();
    }
    CSTAMP(3);
    // ---- trsm of the rows below, 64 rows per step, round-robin over the workgroups ----------------------------
    {
      const int64_t first = (int64_t)(p + 1) * HBO_TILE;
      const int ngroups = (int)(((int64_t)(nblk + 1) * HBO_TILE - first) / 64);
      bool staged = false;
      __syncthreads();
      for (int gidx = wg; gidx < ngroups; gidx += nwg) {
        hbo_chol::trsm_body<T, false>(t, p, first + (int64_t)gidx * 64, smem, !staged);
        staged = true;
      }
    }
    CSTAMP(4);
    grid_barrier(sy, nwg, gen);
    CSTAMP(5);
    // ---- a finished group releases its bulk update ------------------------------------------------------------
    if ((p == g0 + a.q - 1 || p == nblk - 1) && wg == 0 && threadIdx.x == 0) atomicAdd(a.s_panels, 1ull);
  }
  // (after a time-out every group has still been signalled: the host streams are never left waiting)
  if (wg == 0 && threadIdx.x == 0 && sy->timed_out) atomicMin(a.info, 0);   // reported as "not positive definite"
}

}  // namespace

#ifdef HBO_CHAIN_TIMING
extern "C" void hbo_dbg_chain_stamps(unsigned long long* host) { hipMemcpyFromSymbol(host, HIP_SYMBOL(hbo_dbg_chain), sizeof(unsigned long long) * 6 * 256); }
#endif
int chain_lds_bytes() { return 100 * 1024; }
void launch_chain(int dtype, const ChainArgs& a, int nwg, hipStream_t st) {
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&chain_kernel<double>), hipFuncAttributeMaxDynamicSharedMemorySize, chain_lds_bytes());
    hipFuncSetAttribute(reinterpret_cast<const void*>(&chain_kernel<float>), hipFuncAttributeMaxDynamicSharedMemorySize, chain_lds_bytes());
    attr_set = true;
  }
  if (dtype == HBO_F64) hipLaunchKernelGGL((chain_kernel<double>), dim3(nwg), dim3(256), chain_lds_bytes(), st, a);
  else hipLaunchKernelGGL((chain_kernel<float>), dim3(nwg), dim3(256), chain_lds_bytes(), st, a);
}
