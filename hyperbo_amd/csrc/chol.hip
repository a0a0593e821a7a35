// Diagonal-panel kernels of the blocked right-looking Cholesky on gfx950:
//   potf2_kernel  : LDS-resident 128x128 factorisation of the diagonal block (one workgroup),
//                   plus the inverses of its eight 16x16 leaf blocks (written to W).
//   trsm_kernel   : X = A_panel * L_pp^-T by blocked forward substitution on MFMA, one wave per
//                   16 panel rows, L_pp and the leaf inverses staged in LDS as packed 16x16 tiles.
//                   With IDENT=true the same code produces W_pp = L_pp^-1 (trtri base case).
// Replaces what jax.scipy.linalg.cholesky lowers to (hyperbo/basics/linalg.py:31,134).
#include "hbo_internal.h"
#include <limits.h>

namespace {

template <typename T> struct Mma;
template <> struct Mma<double> {
  typedef double acc_t __attribute__((ext_vector_type(4)));
  static __device__ __forceinline__ acc_t mma(double a, double b, acc_t c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ int crow(int lane, int r) { return (lane >> 4) + 4 * r; }
};
template <> struct Mma<float> {
  typedef float acc_t __attribute__((ext_vector_type(4)));
  static __device__ __forceinline__ acc_t mma(float a, float b, acc_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ int crow(int lane, int r) { return 4 * (lane >> 4) + r; }
};

constexpr int NB = HBO_TILE;   // 128
constexpr int LS = NB + 1;     // LDS row stride of the potf2 block (odd -> conflict-free columns)

__device__ __forceinline__ double readlane_t(double v, int lane) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_readlane(lo, lane);
  hi = __builtin_amdgcn_readlane(hi, lane);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ float readlane_t(float v, int lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

// LDS-resident factorisation of one 128x128 diagonal block, blocked by 16 columns:
//   (A) wave 0 factors the 16x16 leaf in registers (lane = row, cross-lane broadcasts via readlane),
//   (B) one thread per remaining row solves its 16 unknowns against the leaf (forward substitution),
//   (C) all four waves apply the rank-16 update to the trailing tiles with MFMA.
// Finally the eight leaf inverses are written to W (used by trsm_kernel and the trtri base case).
template <typename T>
__global__ __launch_bounds__(256) void potf2_kernel(const TaskDesc* tasks, int p, int* info) {
  typedef typename Mma<T>::acc_t acc_t;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  T* sL = reinterpret_cast<T*>(smem);       // [128][129]
  T* sDinv = sL + NB * LS;                  // [128] 1 / diag(L)
  const TaskDesc& t = tasks[blockIdx.x];
  if (p >= t.nblk) return;
  const int64_t ld = t.ld;
  T* Ab = static_cast<T*>(t.A) + (int64_t)p * NB * ld + (int64_t)p * NB;
  T* Wb = static_cast<T*>(t.W) + (int64_t)p * NB * ld + (int64_t)p * NB;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, lq = lane >> 4;

  for (int idx = tid; idx < NB * NB; idx += 256) {
    const int r = idx >> 7, c = idx & 127;
    sL[r * LS + c] = (c <= r) ? Ab[(int64_t)r * ld + c] : (T)0;
  }
  __syncthreads();

  for (int jb = 0; jb < 8; ++jb) {
    const int o = jb * 16;
    // ---- (A) leaf Cholesky, wave 0, lanes 0..15 hold one row each --------------------------
    if (wave == 0) {
      T a[16];
      const int row = o + l15;
#pragma unroll
      for (int c = 0; c < 16; ++c) a[c] = (lane < 16 && c <= l15) ? sL[row * LS + o + c] : (T)0;
      T myinv = (T)0;
      int bad_col = -1;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        T d = readlane_t(a[j], j);
        if (!(d > (T)0)) {   // not PD (or NaN): propagate NaN like jax.scipy.linalg.cholesky
          if (bad_col < 0) bad_col = j;
          d = (T)NAN;
        }
        const T inv = rsqrt(d);
        if (l15 == j) { a[j] = d * inv; myinv = inv; }
        else a[j] = a[j] * inv;
#pragma unroll
        for (int c = j + 1; c < 16; ++c) {
          const T lcj = readlane_t(a[j], c);
          a[c] -= a[j] * lcj;
        }
      }
      if (bad_col >= 0 && lane == 0) atomicMin(&info[blockIdx.x], p * NB + o + bad_col + 1);
      if (lane < 16) {
#pragma unroll
        for (int c = 0; c < 16; ++c)
          if (c <= l15) sL[row * LS + o + c] = a[c];
        sDinv[row] = myinv;
      }
    }
    __syncthreads();
    // ---- (B) panel rows below the leaf: x L_leaf^T = a -------------------------------------
    if (tid < NB && tid >= o + 16) {
      T x[16];
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        T s = sL[tid * LS + o + c];
#pragma unroll
        for (int k = 0; k < 16; ++k)
          if (k < c) s -= x[k] * sL[(o + c) * LS + o + k];
        x[c] = s * sDinv[o + c];
      }
#pragma unroll
      for (int c = 0; c < 16; ++c) sL[tid * LS + o + c] = x[c];
    }
    __syncthreads();
    // ---- (C) trailing update C[I][J] -= X_I X_J^T on MFMA -----------------------------------
    const int m = 7 - jb;               // remaining 16-blocks
    const int ntiles = m * (m + 1) / 2;
    for (int tix = wave; tix < ntiles; tix += 4) {
      int ii = 0;
      while ((ii + 1) * (ii + 2) / 2 <= tix) ++ii;
      const int jj = tix - ii * (ii + 1) / 2;
      const int I = jb + 1 + ii, J = jb + 1 + jj;
      acc_t acc;
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[r] = sL[(I * 16 + Mma<T>::crow(lane, r)) * LS + J * 16 + l15];
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const T af = -sL[(I * 16 + l15) * LS + o + kk * 4 + lq];
        const T bf = sL[(J * 16 + l15) * LS + o + kk * 4 + lq];
        acc = Mma<T>::mma(af, bf, acc);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) sL[(I * 16 + Mma<T>::crow(lane, r)) * LS + J * 16 + l15] = acc[r];
    }
    __syncthreads();
  }

  // write L (lower, zeros above the diagonal inside the block)
  for (int idx = tid; idx < NB * NB; idx += 256) {
    const int rr = idx >> 7, c = idx & 127;
    Ab[(int64_t)rr * ld + c] = (c <= rr) ? sL[rr * LS + c] : (T)0;
  }
  // inverses of the eight 16x16 diagonal leaves: thread = (leaf b, column c), forward substitution
  if (tid < NB) {
    const int b = tid >> 4, c = tid & 15;
    T w[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      T s = (i == c) ? (T)1 : (T)0;
#pragma unroll
      for (int k = 0; k < 16; ++k)
        if (k < i) s -= sL[(b * 16 + i) * LS + b * 16 + k] * w[k];
      w[i] = s * sDinv[b * 16 + i];
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) Wb[(int64_t)(b * 16 + i) * ld + b * 16 + c] = w[i];
  }
}

// packed 16x16 tiles, row stride 17
constexpr int TS = 17;
constexpr int TILE_ELEMS = 16 * TS;                 // 272
__device__ __forceinline__ int tri_index(int I, int J) { return I * (I + 1) / 2 + J; }

template <typename T>
constexpr int trsm_lds_bytes() { return (36 + 8 + 4) * TILE_ELEMS * (int)sizeof(T); }

// grid.x = 64-row group, grid.y = (IDENT ? diagonal block p : unused), grid.z = task
template <typename T, bool IDENT>
__global__ __launch_bounds__(256) void trsm_kernel(const TaskDesc* tasks, int p_arg) {
  typedef typename Mma<T>::acc_t acc_t;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  T* sLt = reinterpret_cast<T*>(smem);          // 36 packed lower tiles of L_pp
  T* sWi = sLt + 36 * TILE_ELEMS;               // 8 leaf inverses
  T* sSc = sWi + 8 * TILE_ELEMS;                // 4 per-wave scratch tiles
  const TaskDesc& t = tasks[blockIdx.z];
  const int p = IDENT ? (int)blockIdx.y : p_arg;
  if (p >= t.nblk) return;
  const int64_t ld = t.ld;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, lq = lane >> 4;
  // first row handled by this workgroup
  int64_t row0;
  if (IDENT) {
    row0 = (int64_t)p * NB + (int64_t)blockIdx.x * 64;            // blockIdx.x in {0,1}
  } else {
    const int64_t first = (int64_t)(p + 1) * NB;
    const int64_t nrows = (int64_t)(t.nblk + 1) * NB - first;     // incl. augmented tile-row
    if ((int64_t)blockIdx.x * 64 >= nrows) return;
    row0 = first + (int64_t)blockIdx.x * 64;
  }
  const T* Lb = static_cast<const T*>(t.A) + (int64_t)p * NB * ld + (int64_t)p * NB;
  const T* Wb = static_cast<const T*>(t.W) + (int64_t)p * NB * ld + (int64_t)p * NB;

  // stage L_pp (lower tiles) and the leaf inverses
  for (int idx = tid; idx < 36 * 256; idx += 256) {
    const int tile = idx >> 8, e = idx & 255, i = e >> 4, j = e & 15;
    // tile -> (I,J)
    int I = 0;
    while ((I + 1) * (I + 2) / 2 <= tile) ++I;
    const int J = tile - I * (I + 1) / 2;
    sLt[tile * TILE_ELEMS + i * TS + j] = Lb[(int64_t)(I * 16 + i) * ld + J * 16 + j];
  }
  for (int idx = tid; idx < 8 * 256; idx += 256) {
    const int b = idx >> 8, e = idx & 255, i = e >> 4, j = e & 15;
    sWi[b * TILE_ELEMS + i * TS + j] = Wb[(int64_t)(b * 16 + i) * ld + b * 16 + j];
  }
  __syncthreads();

  T* sc = sSc + wave * TILE_ELEMS;
  const int64_t wrow0 = row0 + wave * 16;                 // this wave's 16 rows
  const int rg = IDENT ? (int)((wrow0 - (int64_t)p * NB) >> 4) : 0;  // row group inside the block
  T* Ap = IDENT ? nullptr : static_cast<T*>(t.A) + wrow0 * ld + (int64_t)p * NB;
  T* Wout = IDENT ? static_cast<T*>(t.W) + (int64_t)p * NB * ld + (int64_t)p * NB : nullptr;

  T xneg[8][4];  // -X in A-operand layout, per 16-column block
#pragma unroll
  for (int jb = 0; jb < 8; ++jb) {
    acc_t acc;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = Mma<T>::crow(lane, r);
      if (IDENT) acc[r] = (rg == jb && row == l15) ? (T)1 : (T)0;
      else acc[r] = Ap[(int64_t)row * ld + jb * 16 + l15];
    }
    // acc -= sum_{kb<jb} X[:,kb] * L[jb,kb]^T
#pragma unroll
    for (int kb = 0; kb < 8; ++kb) {
      if (kb < jb) {
        const T* lt = sLt + tri_index(jb, kb) * TILE_ELEMS;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          const T bfrag = lt[l15 * TS + kk * 4 + lq];   // B[k][j] = L[jb*16+j][kb*16+k]
          acc = Mma<T>::mma(xneg[kb][kk], bfrag, acc);
        }
      }
    }
    // T (C layout) -> scratch -> A layout
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) sc[Mma<T>::crow(lane, r) * TS + l15] = acc[r];
    __syncthreads();
    T ta[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) ta[kk] = sc[l15 * TS + kk * 4 + lq];
    // X[:,jb] = T * Winv_jb^T
    acc_t x = (acc_t){0, 0, 0, 0};
    const T* wt = sWi + jb * TILE_ELEMS;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const T bfrag = wt[l15 * TS + kk * 4 + lq];       // B[k][j] = Winv[j][k]
      x = Mma<T>::mma(ta[kk], bfrag, x);
    }
    // store
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = Mma<T>::crow(lane, r);
      if (IDENT) {
        // W = X^T ; the diagonal leaves were already written by potf2
        if (rg != jb) Wout[(int64_t)(jb * 16 + l15) * ld + rg * 16 + row] = x[r];
      } else {
        Ap[(int64_t)row * ld + jb * 16 + l15] = x[r];
      }
    }
    // X (C layout) -> scratch -> negated A layout for later column blocks
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) sc[Mma<T>::crow(lane, r) * TS + l15] = x[r];
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) xneg[jb][kk] = -sc[l15 * TS + kk * 4 + lq];
  }
}

template <typename T>
void set_attrs() {
  static bool done = false;
  if (done) return;
  hipFuncSetAttribute(reinterpret_cast<const void*>(&potf2_kernel<T>),
                      hipFuncAttributeMaxDynamicSharedMemorySize, (NB * LS + NB) * (int)sizeof(T));
  hipFuncSetAttribute(reinterpret_cast<const void*>(&trsm_kernel<T, false>),
                      hipFuncAttributeMaxDynamicSharedMemorySize, trsm_lds_bytes<T>());
  hipFuncSetAttribute(reinterpret_cast<const void*>(&trsm_kernel<T, true>),
                      hipFuncAttributeMaxDynamicSharedMemorySize, trsm_lds_bytes<T>());
  done = true;
}

template <typename T>
void potf2_t(const TaskDesc* tasks, int ntasks, int p, int* info, hipStream_t st) {
  set_attrs<T>();
  hipLaunchKernelGGL((potf2_kernel<T>), dim3(ntasks), dim3(256), (NB * LS + NB) * sizeof(T), st, tasks, p, info);
}
template <typename T>
void trsm_t(const TaskDesc* tasks, int ntasks, int p, int max_nblk, hipStream_t st) {
  set_attrs<T>();
  const int nrows = (max_nblk + 1 - (p + 1)) * NB;
  if (nrows <= 0) return;
  hipLaunchKernelGGL((trsm_kernel<T, false>), dim3(nrows / 64, 1, ntasks), dim3(256), trsm_lds_bytes<T>(), st,
                     tasks, p);
}
template <typename T>
void trtri_diag_t(const TaskDesc* tasks, int ntasks, int max_nblk, hipStream_t st) {
  set_attrs<T>();
  hipLaunchKernelGGL((trsm_kernel<T, true>), dim3(2, max_nblk, ntasks), dim3(256), trsm_lds_bytes<T>(), st,
                     tasks, 0);
}

}  // namespace

void launch_potf2(int dtype, const TaskDesc* tasks, int ntasks, int p, int* info, hipStream_t st) {
  if (dtype == HBO_F64) potf2_t<double>(tasks, ntasks, p, info, st);
  else potf2_t<float>(tasks, ntasks, p, info, st);
}
void launch_trsm(int dtype, const TaskDesc* tasks, int ntasks, int p, int max_nblk, hipStream_t st) {
  if (dtype == HBO_F64) trsm_t<double>(tasks, ntasks, p, max_nblk, st);
  else trsm_t<float>(tasks, ntasks, p, max_nblk, st);
}
void launch_trtri_diag(int dtype, const TaskDesc* tasks, int ntasks, int max_nblk, hipStream_t st) {
  if (dtype == HBO_F64) trtri_diag_t<double>(tasks, ntasks, max_nblk, st);
  else trtri_diag_t<float>(tasks, ntasks, max_nblk, st);
}
