// Persistent panel-chain kernel of the blocked Cholesky (single task, look-ahead mode).
//
// potf2 of the diagonal block and trsm of the rows below it -- the two latency-bound steps of every 128-column panel
// -- run inside ONE kernel that stays resident for the whole factorisation: `nwg` workgroups, each requesting more
// than 3/4 of a CU's LDS so that it owns its CU (no GEMM workgroup of the trailing updates fits beside it; beside
// such workgroups potf2 ran 3x slower and waited for slots, profiles/r01_potrf_chain.md).  The GEMM work -- the
// left-looking update of the next block column, the next group's columns (F1), the bulk update (F2) -- stays in
// host-launched kernels on the other CUs, tied to the chain with signal memory instead of kernel boundaries:
//   s_panels : the chain adds 1 per finished panel      -> hipStreamWaitValue64 releases the kernels that need it
//   s_col    : hipStreamWriteValue64(col_base + p) once block column p is up to date -> the chain spins on it
//              before potf2 of panel p.
#include "hbo_internal.h"
#include <limits.h>
#define HBO_DEVICE_ONLY
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
  const TaskDesc& t = a.tasks[0];
  const int nblk = t.nblk;
  const int nwg = gridDim.x, wg = blockIdx.x;
  ChainSync* sy = reinterpret_cast<ChainSync*>(a.sync);
  unsigned gen = 0;

  for (int p = 0; p < nblk; ++p) {
    CSTAMP(0);
    // block column p has to be up to date (host-launched column update / F1 of the previous group)
    if (p > 0) {
      if (threadIdx.x == 0) {
        const unsigned long long need = a.col_base + (unsigned long long)p;
        HBO_SPIN_WHILE(__hip_atomic_load(a.s_col, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < need &&
                       !__hip_atomic_load(&sy->timed_out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
      }
      __syncthreads();
      __threadfence();
    }
    CSTAMP(1);
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
    // ---- a finished panel releases the kernels that consume it -------------------------------------------------
    if (wg == 0 && threadIdx.x == 0) atomicAdd(a.s_panels, 1ull);
  }
  // (after a time-out every group has still been signalled: the host streams are never left waiting)
  if (wg == 0 && threadIdx.x == 0 && sy->timed_out) atomicMin(a.info, 0);   // reported as "not positive definite"
}

}  // namespace

#ifdef HBO_CHAIN_TIMING
extern "C" void hbo_dbg_chain_stamps(unsigned long long* host) { hipMemcpyFromSymbol(host, HIP_SYMBOL(hbo_dbg_chain), sizeof(unsigned long long) * 6 * 256); }
#endif
int chain_lds_bytes() { return 124 * 1024; }   // nothing with >= 36 KB of LDS fits beside a chain workgroup
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
