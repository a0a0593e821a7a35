// Solve with a GIVEN lower Cholesky factor: x = L^-T L^-1 b by blocked forward and back substitution on the device.
// The array-level counterpart of hyperbo/basics/linalg.py:139-145 (`jsla.cho_solve((cached_cholesky, True), x)`): the caller
// hands the factor in as an array (what the reference's GPCache.chol holds), so nothing is factorised and no inverse is formed;
// O(n^2 m) flops, bound by one pass over the lower triangle per sweep.  64-row blocks:
//   trisolve_diag_kernel   : one wave solves the 64x64 diagonal block for every right-hand side (the pivot value travels by
//                            readlane, no barrier on the serial chain),
//   trisolve_update_kernel : the rows still to be solved subtract the block's contribution, one thread per row (forward:
//                            512 contiguous bytes of its row; backward: column-wise, coalesced across the threads).
#include "hbo_internal.h"

namespace {

constexpr int SB = 64;

template <typename T>
__device__ __forceinline__ T lane_value(T v, int lane);
template <> __device__ __forceinline__ double lane_value<double>(double v, int lane) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_readlane(lo, lane); hi = __builtin_amdgcn_readlane(hi, lane);
  return __hiloint2double(hi, lo);
}
template <> __device__ __forceinline__ float lane_value<float>(float v, int lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

// L: n x n row-major (ld = n), rhs: m x npad (one right-hand side per row); block p = rows [64 p, 64 p + 64)
template <typename T, bool TRANS>
__global__ __launch_bounds__(64) void trisolve_diag_kernel(const T* __restrict__ L, int64_t n, T* __restrict__ rhs, int64_t npad, int m, int p) {
  __shared__ T sL[SB * (SB + 1)];
  const int lane = threadIdx.x;
  const int64_t r0 = (int64_t)p * SB;
  for (int i = 0; i < SB; ++i) {
    const int64_t gr = r0 + i, gc = r0 + lane;
    // rows / columns beyond n: identity
    sL[i * (SB + 1) + lane] = (gr < n && gc < n) ? (lane <= i ? L[gr * n + gc] : (T)0) : (gr == gc ? (T)1 : (T)0);
  }
  __syncthreads();
  for (int c = 0; c < m; ++c) {
    T x = rhs[(int64_t)c * npad + r0 + lane];
    if (!TRANS) {
#pragma unroll 8
      for (int j = 0; j < SB; ++j) {
        const T yj = lane_value<T>(x, j) / sL[j * (SB + 1) + j];
        if (lane == j) x = yj;
        else if (lane > j) x -= sL[lane * (SB + 1) + j] * yj;
      }
    } else {
#pragma unroll 8
      for (int j = SB - 1; j >= 0; --j) {
        const T yj = lane_value<T>(x, j) / sL[j * (SB + 1) + j];
        if (lane == j) x = yj;
        else if (lane < j) x -= sL[j * (SB + 1) + lane] * yj;
      }
    }
    rhs[(int64_t)c * npad + r0 + lane] = x;
  }
}

// forward: rows r >= 64 (p + 1):  rhs[r] -= sum_j L[r][64 p + j] y[j];   backward: rows r < 64 p:  rhs[r] -= sum_j L[64 p + j][r] y[j]
template <typename T, bool TRANS>
__global__ __launch_bounds__(256) void trisolve_update_kernel(const T* __restrict__ L, int64_t n, T* __restrict__ rhs, int64_t npad, int m, int p) {
  __shared__ T sy[SB];
  const int64_t r0 = (int64_t)p * SB;
  const int64_t r = TRANS ? (int64_t)blockIdx.x * 256 + threadIdx.x : r0 + SB + (int64_t)blockIdx.x * 256 + threadIdx.x;
  const bool live = TRANS ? r < r0 : r < n;
  for (int c = 0; c < m; ++c) {
    __syncthreads();
    if (threadIdx.x < SB) sy[threadIdx.x] = rhs[(int64_t)c * npad + r0 + threadIdx.x];
    __syncthreads();
    if (!live) continue;
    T acc = 0;
    if (!TRANS) {
      const T* row = L + r * n + r0;
#pragma unroll 8
      for (int j = 0; j < SB; ++j) acc += row[j] * sy[j];
    } else {
      for (int j = 0; j < SB; ++j) if (r0 + j < n) acc += L[(r0 + j) * n + r] * sy[j];
    }
    rhs[(int64_t)c * npad + r] -= acc;
  }
}

template <typename T>
void chol_solve_t(const T* L, int64_t n, T* rhs, int64_t npad, int m, hipStream_t st) {
  const int nb = (int)((n + SB - 1) / SB);
  for (int p = 0; p < nb; ++p) {
    hipLaunchKernelGGL((trisolve_diag_kernel<T, false>), dim3(1), dim3(64), 0, st, L, n, rhs, npad, m, p);
    const int64_t rest = n - (int64_t)(p + 1) * SB;
    if (rest > 0) hipLaunchKernelGGL((trisolve_update_kernel<T, false>), dim3((unsigned)((rest + 255) / 256)), dim3(256), 0, st, L, n, rhs, npad, m, p);
  }
  for (int p = nb - 1; p >= 0; --p) {
    hipLaunchKernelGGL((trisolve_diag_kernel<T, true>), dim3(1), dim3(64), 0, st, L, n, rhs, npad, m, p);
    const int64_t above = (int64_t)p * SB;
    if (above > 0) hipLaunchKernelGGL((trisolve_update_kernel<T, true>), dim3((unsigned)((above + 255) / 256)), dim3(256), 0, st, L, n, rhs, npad, m, p);
  }
}

}  // namespace

// rhs: m x npad on the device (npad >= n rounded up to 64; padding entries are solved against the identity), in place
void launch_chol_solve(int dtype, const void* L, int64_t n, void* rhs, int64_t npad, int m, hipStream_t st) {
  if (dtype == HBO_F64) chol_solve_t<double>(static_cast<const double*>(L), n, static_cast<double*>(rhs), npad, m, st);
  else chol_solve_t<float>(static_cast<const float*>(L), n, static_cast<float*>(rhs), npad, m, st);
}
