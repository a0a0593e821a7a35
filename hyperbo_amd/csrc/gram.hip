// HBM-bound / elementwise kernels of the GP hot path on gfx950: pairwise-kernel Gram build
// (LDS-tiled X blocks, coalesced 16-byte stores, fused +(sigma^2+eps) I), MLP features,
// mean / augmented residual rows, NLL reduction, s = W^T z, gradient contraction
// sum_ij G_ij dK_ij/dtheta (K recomputed on the fly, K^-1 read once), posterior epilogue with
// fused EI / PI / UCB.
//
// Reference restated: hyperbo/gp_utils/kernel.py:29-145 (Gram), basis_functions.py:24-36 (MLP),
// mean.py:30-79, basics/linalg.py:36-69 (jitter), objectives.py:144-156 (NLL),
// gp.py:242-305 (posterior), bo_utils/acfun.py:96-142 (acquisition).
#include "hbo_internal.h"
#include <limits.h>
#include <math.h>

namespace {

template <typename T> struct V16;
template <> struct V16<double> { typedef double type __attribute__((ext_vector_type(2))); };
template <> struct V16<float> { typedef float type __attribute__((ext_vector_type(4))); };

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
// sum over a 256-thread block; sred must hold 4 doubles. Result valid in every thread.
__device__ __forceinline__ double block_sum(double v, double* sred) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sred[threadIdx.x >> 6] = v;
  __syncthreads();
  return sred[0] + sred[1] + sred[2] + sred[3];
}

template <typename T>
__device__ __forceinline__ T kfun(int kid, T acc, T sv, T inv_sigma2, T bias2) {
  switch (kid) {
    case HBO_KERNEL_SE: return sv * exp((T)-0.5 * acc);
    case HBO_KERNEL_MATERN32: { T r = sqrt((T)3 * acc); return sv * ((T)1 + r) * exp(-r); }
    case HBO_KERNEL_MATERN52: { T r = sqrt((T)5 * acc); return sv * ((T)1 + r + r * r / (T)3) * exp(-r); }
    default: return acc * inv_sigma2 + bias2;
  }
}
// d k / d u (u = scaled squared distance); 0 where u == 0 for Matern (linalg.py:183-188)
template <typename T>
__device__ __forceinline__ T dk_du(int kid, T u, T k, T sv) {
  switch (kid) {
    case HBO_KERNEL_SE: return (T)-0.5 * k;
    case HBO_KERNEL_MATERN32: { T r = sqrt((T)3 * u); return u == (T)0 ? (T)0 : -sv * (T)1.5 * exp(-r); }
    case HBO_KERNEL_MATERN52: { T r = sqrt((T)5 * u); return u == (T)0 ? (T)0 : -sv * ((T)5 / (T)6) * exp(-r) * ((T)1 + r); }
    default: return (T)0;
  }
}

constexpr int DC = 16;     // feature chunk staged in LDS
constexpr int SXS = 132;   // LDS row stride of a staged [DC][128] block

// stage rows [r0, r0+128) x features [d0, d0+DC) of x (n x fdim) into s[dd][row], scaled
template <typename T, int NQ = 8>
__device__ __forceinline__ void stage_x(T* s, const T* __restrict__ x, int64_t n, int fdim, int64_t r0,
                                        int d0, const double* inv_ls, bool scale, int tid) {
  const int dd = tid & 15, rr0 = tid >> 4;
  const int d = d0 + dd;
  const T sc = (scale && d < fdim) ? (T)inv_ls[d] : (T)1;
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int rr = rr0 + 16 * q;
    const int64_t row = r0 + rr;
    T v = (T)0;
    if (row < n && d < fdim) v = gld(x + row * fdim + d) * sc;
    s[dd * SXS + rr] = v;
  }
}

// tile = 64 rows x 128 columns per 256-thread workgroup, 4x8 register micro-tile per thread
// (an 8x8 micro-tile needs 256 VGPRs in fp64 -> 1 wave/SIMD and exposed exp/store latency).
constexpr int GRA = 4;            // rows per thread
constexpr int GTR = 16 * GRA;     // tile rows

// KID: covariance id as a compile-time constant -- with a run-time id every one of a thread's 32 elements carried the
// switch over all four covariances (188 VGPRs, 2 waves per SIMD, constants re-materialised per exponential)
template <typename T, bool PADDED, int KID>
__global__ __launch_bounds__(256) void gram_kernel(GramArgs g, const ModelDev* __restrict__ md) {
  typedef typename V16<T>::type vec_t;
  constexpr int VEC = 16 / sizeof(T);
  __shared__ T sA[DC * SXS];
  __shared__ T sB[DC * SXS];
  // ti in units of GTR rows, tj in units of 128 columns.  Symmetric mode computes the tiles with tj <= ti/2 only, and
  // workgroup i runs on XCD (i + const) mod 8: with tj = blockIdx.x the low XCDs would get one more tile than the high ones in
  // every row (grid.x is a multiple of 8 for the sizes that matter), so the column index is rotated by the row
  const int ti = blockIdx.y;
  const int tj = g.symmetric ? (int)((blockIdx.x + blockIdx.y) % gridDim.x) : (int)blockIdx.x;
  const T* x1; const T* x2; T* out; int64_t n1, n2, ldo; int64_t e1, e2;  // e*: padded extents
  if (g.tasks) {
    const TaskDesc& t = g.tasks[blockIdx.z];
    if ((int64_t)ti * GTR >= t.npad || tj >= t.nblk) return;
    x1 = x2 = static_cast<const T*>(t.F);
    out = static_cast<T*>(t.A);
    n1 = n2 = t.n; ldo = t.ld; e1 = e2 = t.npad;
  } else {
    x1 = static_cast<const T*>(g.x1); x2 = static_cast<const T*>(g.x2); out = static_cast<T*>(g.out);
    n1 = g.n1; n2 = g.n2; ldo = g.ldo; e1 = PADDED ? g.n1pad : g.n1; e2 = PADDED ? g.n2pad : g.n2;
  }
  const int64_t r0 = (int64_t)ti * GTR, c0 = (int64_t)tj * HBO_TILE;
  if (g.symmetric && c0 > r0 + GTR - 1) return;   // entirely above the diagonal
  const int fdim = g.fdim;
  constexpr int kid = KID;
  constexpr bool is_dot = (kid == HBO_KERNEL_DOT);
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;

  T acc[GRA][8];
#pragma unroll
  for (int a = 0; a < GRA; ++a)
#pragma unroll
    for (int b = 0; b < 8; ++b) acc[a][b] = (T)0;

  for (int d0 = 0; d0 < fdim; d0 += DC) {
    __syncthreads();
    stage_x<T, GRA>(sA, x1, n1, fdim, r0, d0, md->inv_ls, !is_dot, tid);
    stage_x<T, 8>(sB, x2, n2, fdim, c0, d0, md->inv_ls, !is_dot, tid);
    __syncthreads();
    const int dlim = (fdim - d0) < DC ? (fdim - d0) : DC;
    for (int dd = 0; dd < dlim; ++dd) {
      T av[GRA], bv[8];
#pragma unroll
      for (int a = 0; a < GRA; ++a) av[a] = sA[dd * SXS + ty + 16 * a];
#pragma unroll
      for (int q = 0; q < 8; ++q) bv[q] = sB[dd * SXS + 16 * VEC * (q / VEC) + VEC * tx + (q % VEC)];
      if (is_dot) {
#pragma unroll
        for (int a = 0; a < GRA; ++a)
#pragma unroll
          for (int q = 0; q < 8; ++q) acc[a][q] += av[a] * bv[q];
      } else {
#pragma unroll
        for (int a = 0; a < GRA; ++a)
#pragma unroll
          for (int q = 0; q < 8; ++q) { const T df = av[a] - bv[q]; acc[a][q] += df * df; }
      }
    }
  }

  const T sv = (T)md->sv;
  const T inv_sigma2 = (T)(1.0 / (md->dot_sigma * md->dot_sigma));
  const T bias2 = (T)(md->dot_bias * md->dot_bias);
  const T diag_add = (T)(md->noise + md->eps);
#pragma unroll
  for (int a = 0; a < GRA; ++a) {
    const int64_t row = r0 + ty + 16 * a;
    if (row >= e1) continue;
#pragma unroll
    for (int qb = 0; qb < 8 / VEC; ++qb) {
      const int64_t col0 = c0 + 16 * VEC * qb + VEC * tx;
      T vals[VEC];
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        const int64_t col = col0 + e;
        T v;
        if (row < n1 && col < n2) {
#ifdef HBO_GRAM_NOEXP
          v = acc[a][qb * VEC + e] * sv;
#else
          v = kfun<T>(kid, acc[a][qb * VEC + e], sv, inv_sigma2, bias2);
#endif
          if (g.symmetric && row == col) v += diag_add;
        } else {
          v = (g.symmetric && row == col) ? (T)1 : (T)0;   // identity / zero padding
        }
        vals[e] = v;
      }
      if (PADDED) {
        vec_t vv;
#pragma unroll
        for (int e = 0; e < VEC; ++e) vv[e] = vals[e];
#ifdef HBO_GRAM_NOSTORE
        if (vv[0] == (T)123.456) gst(reinterpret_cast<vec_t*>(out + row * ldo + col0), vv);
#else
        gst(reinterpret_cast<vec_t*>(out + row * ldo + col0), vv);
#endif
      } else {
#pragma unroll
        for (int e = 0; e < VEC; ++e)
          if (col0 + e < e2) gst(out + row * ldo + col0 + e, vals[e]);
      }
    }
  }
}

template <typename T>
__global__ void kdiag_kernel(const T* __restrict__ f, int64_t n, int fdim, const ModelDev* __restrict__ md,
                             T* out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (md->kernel_id == HBO_KERNEL_DOT) {
    T s = 0;
    for (int d = 0; d < fdim; ++d) { const T v = f[i * fdim + d]; s += v * v; }
    out[i] = s * (T)(1.0 / (md->dot_sigma * md->dot_sigma)) + (T)(md->dot_bias * md->dot_bias);
  } else {
    out[i] = (T)md->sv;
  }
}

// out[i][o] = tanh(sum_k in[i][k] w[k][o] + b[o])   (flax Dense + tanh)
template <typename T>
__global__ void dense_tanh_kernel(const T* __restrict__ in, const T* __restrict__ w, const T* __restrict__ b,
                                  T* __restrict__ out, int64_t n, int fin, int fout) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n * fout) return;
  const int64_t i = idx / fout;
  const int o = (int)(idx % fout);
  T s = b[o];
  for (int k = 0; k < fin; ++k) s += in[i * fin + k] * w[(int64_t)k * fout + o];
  out[idx] = tanh(s);
}

template <typename T>
__device__ __forceinline__ T mean_at(const ModelDev* md, const T* fm, int fmean, int64_t i) {
  switch (md->mean_id) {
    case HBO_MEAN_ZERO: return (T)0;
    case HBO_MEAN_CONSTANT: return (T)md->constant;
    default: {
      T s = (T)md->linear_bias;
      for (int d = 0; d < fmean; ++d) s += fm[i * fmean + d] * (T)md->lin_w[d];
      return s;
    }
  }
}

template <typename T>
__global__ void mean_kernel(const T* __restrict__ fm, int64_t n, int fmean, const ModelDev* __restrict__ md,
                            T* out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  out[i] = mean_at<T>(md, fm, fmean, i);
}

// augmented tile-row: row b < naug holds  aug_src[b*n + j] + e_b * mu_j  (see TaskDesc); everything else zero.
template <typename T>
__global__ void aug_rows_kernel(const TaskDesc* tasks, const ModelDev* __restrict__ md) {
  const TaskDesc& t = tasks[blockIdx.z];
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= t.npad) return;
  T* Ar = static_cast<T*>(t.A) + (int64_t)t.npad * t.ld + j;
  const T* ys = static_cast<const T*>(t.ysum);
  T mu = (T)0;
  if (j < t.n) mu = mean_at<T>(md, static_cast<const T*>(t.Fm), t.fmean, j);
  for (int a = 0; a < HBO_TILE; ++a) {
    T v = (T)0;
    if (a < t.naug && j < t.n) {
      const T e = (T)(t.e_all + (a == t.naug - 1 ? t.e_last : 0.0));
      v = ys[(int64_t)a * t.n + j] + e * mu;
    }
    Ar[(int64_t)a * t.ld] = v;
  }
}

// f_t = c * sum_b |z_b|^2 + 2 lh * sum log diag L + const   (NLL: objectives.py:153-155; EKL: utils.py:84-106)
template <typename T>
__global__ __launch_bounds__(256) void nll_reduce_kernel(const TaskDesc* tasks, const int* info, double* out) {
  __shared__ double sred[4];
  const TaskDesc& t = tasks[blockIdx.x];
  const T* A = static_cast<const T*>(t.A);
  double ld_sum = 0, q = 0;
  // four elements per thread and pass: the strided diagonal loads of a pass are in flight together (this kernel, the
  // dmu and the finalize kernel sit serially at the end of an evaluation: 36 + 25 + 55 us before)
  for (int64_t i0 = threadIdx.x; i0 < t.n; i0 += 1024) {
    T dg[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { const int64_t i = i0 + 256 * u; dg[u] = i < t.n ? A[i * t.ld + i] : (T)1; }
    for (int b = 0; b < t.naug; ++b) {
      const T* zr = A + ((int64_t)t.npad + b) * t.ld;
#pragma unroll
      for (int u = 0; u < 4; ++u) { const int64_t i = i0 + 256 * u; const double z = i < t.n ? (double)zr[i] : 0.0; q += z * z; }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) ld_sum += log((double)dg[u]);
  }
  ld_sum = block_sum(ld_sum, sred);
  q = block_sum(q, sred);
  if (threadIdx.x == 0) {
    double v = t.coef_c * q + 2.0 * t.coef_lh * ld_sum + t.coef_const;
    if (info[blockIdx.x] != 0x7fffffff) v = NAN;
    out[blockIdx.x] = v;
  }
}

// s = W^T z, stage 1: partial[rc][col] over 512-row chunks, stored in the scratch matrix S.
template <typename T>
__global__ __launch_bounds__(256) void wtz_partial_kernel(const TaskDesc* tasks, int aug_row, const T* xover) {
  __shared__ T sred[256];
  const TaskDesc& t = tasks[blockIdx.z];
  const int cb = blockIdx.x, rc = blockIdx.y;
  if (cb >= t.nblk || (!xover && aug_row >= t.naug)) return;
  const int64_t row_lo = (int64_t)rc * 512;
  if (row_lo >= t.npad) return;
  T* part = static_cast<T*>(t.wscr) + (int64_t)rc * t.ld + (int64_t)cb * HBO_TILE;
  const int col = threadIdx.x & 127, half = threadIdx.x >> 7;
  T acc = (T)0;
  if (row_lo + 512 > (int64_t)cb * HBO_TILE) {  // chunk reaches the lower triangle
    const T* W = static_cast<const T*>(t.W);
    const T* z = xover ? xover : static_cast<const T*>(t.A) + ((int64_t)t.npad + aug_row) * t.ld;
    int64_t r_begin = row_lo + half * 256, r_end = r_begin + 256;
    if (r_end > t.npad) r_end = t.npad;
    const int64_t diag0 = (int64_t)cb * HBO_TILE;
    if (r_begin < diag0) r_begin = diag0;
    // eight independent accumulators: the loads of a row group are in flight together (a single dependent chain ran
    // at 1.9 TB/s and took 31 us on one 128-row block; four: 2.7 TB/s)
    T a[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) a[u] = (T)0;
    const T* wp = W + diag0 + col;
    int64_t r = r_begin;
    for (; r + 8 <= r_end; r += 8) {
      T w[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) w[u] = wp[(r + u) * t.ld];
#pragma unroll
      for (int u = 0; u < 8; ++u) a[u] += w[u] * z[r + u];
    }
    for (; r < r_end; ++r) a[0] += wp[r * t.ld] * z[r];
    acc = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
  }
  sred[threadIdx.x] = acc;
  __syncthreads();
  if (half == 0) part[col] = sred[col] + sred[col + 128];
}
template <typename T>
__global__ void wtz_final_kernel(const TaskDesc* tasks, int out_col, int out_ld, T* oover) {
  const TaskDesc& t = tasks[blockIdx.z];
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= t.npad || (!oover && out_col >= t.naug)) return;
  const T* part = static_cast<const T*>(t.wscr);
  const int nrc = (t.npad + 511) / 512;
  T s = (T)0;
  for (int rc = 0; rc < nrc; ++rc) s += part[(int64_t)rc * t.ld + j];
  if (oover) oover[j] = s;
  else static_cast<T*>(t.svec)[(int64_t)out_col * t.npad + j] = s;   // per-task stride (ragged tasks)
}

// ---------------------------------------------------------------------------------------
// gradient contraction over the lower tiles, G = d objective / d K1:
//   NLL / EKL : G_ij = lh Kinv_ij - c sum_b alpha_b,i alpha_b,j   (S = K1^-1, alpha_b = K1^-1 row_b in svec)
//   EUC       : G_ij = K1_ij - sum_{b<m} V_b,i V_b,j              (= K1 - C0, un-normalised; V = augmented rows;
//               the 1/|C0-K1|_F factor is applied by grad_finalize from the Frobenius accumulator)
// partial sums of G_ij * dK_ij/dtheta per tile.
// accumulators: SE/Matern: [0] sum G K, [1] tr G, [2+d] sum G dk/du ds_d^2, [2+fdim] sum G^2
//               dot      : [0] sum G <fi,fj>, [1] tr G, [2] sum G,          [3] sum G^2
// ---------------------------------------------------------------------------------------
// outer-product vectors of a task: (pointer, row stride, count)
template <typename T>
__device__ __forceinline__ const T* outer_vecs(const TaskDesc& t, int obj, int64_t& stride, int& count) {
  if (obj == OBJ_EUC) {
    stride = t.ld; count = t.naug - 1;
    return static_cast<const T*>(t.A) + (int64_t)t.npad * t.ld;
  }
  stride = t.npad; count = t.naug;
  return static_cast<const T*>(t.svec);
}
// MULTI = false: the NLL fast path (one outer-product vector, no Frobenius accumulator)
template <typename T, bool MULTI, int KID>
__global__ __launch_bounds__(256) void grad_contract_kernel(const TaskDesc* tasks, const ModelDev* __restrict__ md,
                                                            int fdim, int nacc, int obj_arg, double* partials,
                                                            int64_t stride_task) {
  const int obj = MULTI ? obj_arg : (int)OBJ_NLL;
  __shared__ T sA[DC * SXS];
  __shared__ T sB[DC * SXS];
  __shared__ double swred[4][DC + 4];   // per-wave partial sums: 4 scalars + one per staged feature
  // 64 x 128 half tiles (blockIdx.x counts 64-row units): a 4 x 8 register micro-tile per thread instead of 8 x 8 keeps
  // the kernel at ~130 VGPRs (3 waves per SIMD instead of 2: it is bound by the latency of its fp64 exponentials)
  const TaskDesc& t = tasks[blockIdx.z];
  const int th = blockIdx.x, ti = th >> 1, tj = blockIdx.y;
  if (ti >= t.nblk || tj > ti) return;
  constexpr int VEC = 16 / sizeof(T);
  constexpr int kid = KID;   // compile-time covariance id, as in gram_kernel
  constexpr bool is_dot = (kid == HBO_KERNEL_DOT);
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  const int64_t r0 = (int64_t)th * GTR, c0 = (int64_t)tj * HBO_TILE;
  const T* F = static_cast<const T*>(t.F);
  const T* S = static_cast<const T*>(t.S);
  int64_t vstride; int nvec_rt;
  const T* sv_ = outer_vecs<T>(t, obj, vstride, nvec_rt);
  const int nvec = MULTI ? nvec_rt : 1;
  const bool euc = MULTI && (obj == OBJ_EUC);
  const int64_t n = t.n;
  double* out = partials + (int64_t)blockIdx.z * stride_task + (((int64_t)ti * (ti + 1) / 2 + tj) * 2 + (th & 1)) * nacc;

  T acc[GRA][8];
#pragma unroll
  for (int a = 0; a < GRA; ++a)
#pragma unroll
    for (int b = 0; b < 8; ++b) acc[a][b] = (T)0;
  for (int d0 = 0; d0 < fdim; d0 += DC) {
    __syncthreads();
    stage_x<T, GRA>(sA, F, n, fdim, r0, d0, md->inv_ls, !is_dot, tid);
    stage_x<T>(sB, F, n, fdim, c0, d0, md->inv_ls, !is_dot, tid);
    __syncthreads();
    const int dlim = (fdim - d0) < DC ? (fdim - d0) : DC;
    for (int dd = 0; dd < dlim; ++dd) {
      T av[GRA], bv[8];
#pragma unroll
      for (int a = 0; a < GRA; ++a) av[a] = sA[dd * SXS + ty + 16 * a];
#pragma unroll
      for (int q = 0; q < 8; ++q) bv[q] = sB[dd * SXS + 16 * VEC * (q / VEC) + VEC * tx + (q % VEC)];
#pragma unroll
      for (int a = 0; a < GRA; ++a)
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          if (is_dot) acc[a][q] += av[a] * bv[q];
          else { const T df = av[a] - bv[q]; acc[a][q] += df * df; }
        }
    }
  }
  const T sv = (T)md->sv;
  const T inv_sigma2 = (T)(1.0 / (md->dot_sigma * md->dot_sigma));
  const T bias2 = (T)(md->dot_bias * md->dot_bias);
  const T lh = (T)t.coef_lh, cc = (T)t.coef_c, noise = (T)md->noise;
  const T wt = (ti == tj) ? (T)1 : (T)2;   // off-diagonal tiles stand for their mirror image too
  double a_gk = 0, a_tr = 0, a_g = 0, a_fro = 0;
  // gw[a][q] = weight * G_ij * dk/du  (re-uses acc storage)
  typedef typename V16<T>::type vec_t;
  // vector b = 0 for this thread's 8 columns (vectors are zero-padded to npad, S has full padded tiles)
  T sj[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) sj[q] = (T)0;
  if (nvec > 0) {
#pragma unroll
    for (int qb = 0; qb < 8 / VEC; ++qb) {
      const vec_t v = gld(reinterpret_cast<const vec_t*>(sv_ + c0 + 16 * VEC * qb + VEC * tx));
#pragma unroll
      for (int e = 0; e < VEC; ++e) sj[qb * VEC + e] = v[e];
    }
  }
#pragma unroll
  for (int a = 0; a < GRA; ++a) {
    const int64_t row = r0 + ty + 16 * a;
    const T si = nvec > 0 ? gld(sv_ + row) : (T)0;
    T kinv_row[8], outer[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) { outer[q] = si * sj[q]; kinv_row[q] = (T)0; }
    for (int b = 1; b < nvec; ++b) {   // EKL / EUC: further outer-product vectors
      const T* vb = sv_ + (int64_t)b * vstride;
      const T sib = gld(vb + row);
#pragma unroll
      for (int qb = 0; qb < 8 / VEC; ++qb) {
        const vec_t v = gld(reinterpret_cast<const vec_t*>(vb + c0 + 16 * VEC * qb + VEC * tx));
#pragma unroll
        for (int e = 0; e < VEC; ++e) outer[qb * VEC + e] += sib * v[e];
      }
    }
    if (!euc) {
#pragma unroll
      for (int qb = 0; qb < 8 / VEC; ++qb) {
        const vec_t v = gld(reinterpret_cast<const vec_t*>(S + row * t.ld + c0 + 16 * VEC * qb + VEC * tx));
#pragma unroll
        for (int e = 0; e < VEC; ++e) kinv_row[qb * VEC + e] = v[e];
      }
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int64_t col = c0 + 16 * VEC * (q / VEC) + VEC * tx + (q % VEC);
      T gw = (T)0;
      if (row < n && col < n) {
        const T u = acc[a][q];
        const T k = kfun<T>(kid, u, sv, inv_sigma2, bias2);
        const T G0 = euc ? (k + (row == col ? noise : (T)0) - outer[q]) : (lh * kinv_row[q] - cc * outer[q]);
        const T G = G0 * wt;
        if (MULTI) a_fro += (double)(G0 * G);
        if (is_dot) { a_gk += (double)(G * u); a_g += (double)G; }
        else { a_gk += (double)(G * k); gw = G * dk_du<T>(kid, u, k, sv); }
        if (row == col) a_tr += (double)G;
      }
      acc[a][q] = gw;
    }
  }
  // block sums: per-wave sums meet in LDS, one barrier for all accumulators (the tile's 40 us are the 64 fp64
  // exponentials per thread, not the reductions -- measured equal with a barrier pair per accumulator)
  const int lane = tid & 63, wave = tid >> 6;
  a_gk = wave_sum(a_gk); a_tr = wave_sum(a_tr);
  if (is_dot) a_g = wave_sum(a_g);
  if (MULTI) a_fro = wave_sum(a_fro);
  if (lane == 0) { swred[wave][0] = a_gk; swred[wave][1] = a_tr; swred[wave][2] = a_g; swred[wave][3] = a_fro; }
  __syncthreads();
  if (tid == 0) {
    out[0] = (swred[0][0] + swred[1][0]) + (swred[2][0] + swred[3][0]);
    out[1] = (swred[0][1] + swred[1][1]) + (swred[2][1] + swred[3][1]);
    if (is_dot) out[2] = (swred[0][2] + swred[1][2]) + (swred[2][2] + swred[3][2]);
    out[nacc - 1] = (swred[0][3] + swred[1][3]) + (swred[2][3] + swred[3][3]);
  }
  if (is_dot) return;
  // second pass over the features: sum gw * ds_d^2
  for (int d0 = 0; d0 < fdim; d0 += DC) {
    __syncthreads();
    stage_x<T, GRA>(sA, F, n, fdim, r0, d0, md->inv_ls, true, tid);
    stage_x<T>(sB, F, n, fdim, c0, d0, md->inv_ls, true, tid);
    __syncthreads();
    const int dlim = (fdim - d0) < DC ? (fdim - d0) : DC;
    for (int dd = 0; dd < dlim; ++dd) {
      T av[GRA], bv[8];
#pragma unroll
      for (int a = 0; a < GRA; ++a) av[a] = sA[dd * SXS + ty + 16 * a];
#pragma unroll
      for (int q = 0; q < 8; ++q) bv[q] = sB[dd * SXS + 16 * VEC * (q / VEC) + VEC * tx + (q % VEC)];
      T s = (T)0;
#pragma unroll
      for (int a = 0; a < GRA; ++a)
#pragma unroll
        for (int q = 0; q < 8; ++q) { const T df = av[a] - bv[q]; s += acc[a][q] * df * df; }
      const double ws = wave_sum((double)s);
      if (lane == 0) swred[wave][4 + dd] = ws;
    }
    __syncthreads();
    if (tid < dlim) out[2 + d0 + tid] = (swred[0][4 + tid] + swred[1][4 + tid]) + (swred[2][4 + tid] + swred[3][4 + tid]);
  }
}

// ---------------------------------------------------------------------------------------
// d nll / d features for MLP-basis kernels (hyperbo/gp_utils/kernel.py:148-183): per lower tile
//   dF[a][d] += c_d * sum_j g_aj (fs_a - fs_j)_d        (rows of the tile)
//   dF[j][d] -= c_d * sum_a g_aj (fs_a - fs_j)_d        (columns, off-diagonal tiles only)
// with g = G * dk/du, c_d = 4/ls_d (SE / Matern);  dot product: dF[a] += 2/sigma^2 sum_j G_aj f_j.
// Accumulated with fp64 atomics into tasks[t].dF (n x fdim doubles, zeroed by the caller).
// ---------------------------------------------------------------------------------------
template <typename T, int KID>
__global__ __launch_bounds__(256) void grad_feat_kernel(const TaskDesc* tasks, const ModelDev* __restrict__ md,
                                                        int fdim, int obj) {
  __shared__ T sA[DC * SXS];
  __shared__ T sB[DC * SXS];
  const TaskDesc& t = tasks[blockIdx.z];
  const int ti = blockIdx.x, tj = blockIdx.y;
  if (ti >= t.nblk || tj > ti) return;
  constexpr int VEC = 16 / sizeof(T);
  constexpr int kid = KID;
  constexpr bool is_dot = (kid == HBO_KERNEL_DOT);
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  const int64_t r0 = (int64_t)ti * HBO_TILE, c0 = (int64_t)tj * HBO_TILE;
  const T* F = static_cast<const T*>(t.F);
  const T* S = static_cast<const T*>(t.S);
  int64_t vstride; int nvec;
  const T* sv_ = outer_vecs<T>(t, obj, vstride, nvec);
  const bool euc = (obj == OBJ_EUC);
  double* dF = static_cast<double*>(t.dF);
  const int64_t n = t.n;

  T acc[8][8];
#pragma unroll
  for (int a = 0; a < 8; ++a)
#pragma unroll
    for (int b = 0; b < 8; ++b) acc[a][b] = (T)0;
  for (int d0 = 0; d0 < fdim; d0 += DC) {
    __syncthreads();
    stage_x<T>(sA, F, n, fdim, r0, d0, md->inv_ls, !is_dot, tid);
    stage_x<T>(sB, F, n, fdim, c0, d0, md->inv_ls, !is_dot, tid);
    __syncthreads();
    const int dlim = (fdim - d0) < DC ? (fdim - d0) : DC;
    for (int dd = 0; dd < dlim; ++dd) {
      T av[8], bv[8];
#pragma unroll
      for (int a = 0; a < 8; ++a) av[a] = sA[dd * SXS + ty + 16 * a];
#pragma unroll
      for (int q = 0; q < 8; ++q) bv[q] = sB[dd * SXS + 16 * VEC * (q / VEC) + VEC * tx + (q % VEC)];
#pragma unroll
      for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          if (is_dot) acc[a][q] += av[a] * bv[q];
          else { const T df = av[a] - bv[q]; acc[a][q] += df * df; }
        }
    }
  }
  const T sv = (T)md->sv;
  const T inv_sigma2 = (T)(1.0 / (md->dot_sigma * md->dot_sigma));
  const T bias2 = (T)(md->dot_bias * md->dot_bias);
  const T lh = (T)t.coef_lh, cc = (T)t.coef_c, noise = (T)md->noise;
  // g[a][q] = G_ij * dk/du (SE/Matern) or G_ij (dot);  G as in grad_contract_kernel
#pragma unroll
  for (int a = 0; a < 8; ++a) {
    const int64_t row = r0 + ty + 16 * a;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int64_t col = c0 + 16 * VEC * (q / VEC) + VEC * tx + (q % VEC);
      T g = (T)0;
      if (row < n && col < n) {
        const T u = acc[a][q];
        const T k = kfun<T>(kid, u, sv, inv_sigma2, bias2);
        T outer = (T)0;
        for (int b = 0; b < nvec; ++b) outer += sv_[(int64_t)b * vstride + row] * sv_[(int64_t)b * vstride + col];
        const T G = euc ? (k + (row == col ? noise : (T)0) - outer) : (lh * S[row * t.ld + col] - cc * outer);
        if (is_dot) g = G;
        else g = G * dk_du<T>(kid, u, k, sv);
      }
      acc[a][q] = g;
    }
  }
  const bool offdiag = (ti != tj);
  for (int d0 = 0; d0 < fdim; d0 += DC) {
    __syncthreads();
    stage_x<T>(sA, F, n, fdim, r0, d0, md->inv_ls, !is_dot, tid);
    stage_x<T>(sB, F, n, fdim, c0, d0, md->inv_ls, !is_dot, tid);
    __syncthreads();
    const int dlim = (fdim - d0) < DC ? (fdim - d0) : DC;
    for (int dd = 0; dd < dlim; ++dd) {
      const int d = d0 + dd;
      const double cd = is_dot ? 2.0 / (md->dot_sigma * md->dot_sigma) : 4.0 * md->inv_ls[d];
      T av[8], bv[8];
#pragma unroll
      for (int a = 0; a < 8; ++a) av[a] = sA[dd * SXS + ty + 16 * a];
#pragma unroll
      for (int q = 0; q < 8; ++q) bv[q] = sB[dd * SXS + 16 * VEC * (q / VEC) + VEC * tx + (q % VEC)];
      double rs[8], cs[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) cs[q] = 0;
#pragma unroll
      for (int a = 0; a < 8; ++a) {
        double s = 0;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          if (is_dot) { s += (double)(acc[a][q] * bv[q]); cs[q] += (double)(acc[a][q] * av[a]); }
          else { const T w = acc[a][q] * (av[a] - bv[q]); s += (double)w; cs[q] -= (double)w; }
        }
        rs[a] = s;
      }
      // rows: reduce over the 16 tx lanes (consecutive lanes of a wave)
#pragma unroll
      for (int a = 0; a < 8; ++a) {
        double s = rs[a];
        s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); s += __shfl_xor(s, 4); s += __shfl_xor(s, 8);
        const int64_t row = r0 + ty + 16 * a;
        if (tx == 0 && row < n) atomicAdd(&dF[row * fdim + d], cd * s);
      }
      if (offdiag) {
        // columns: reduce over the 4 ty values inside the wave, one atomic per wave and column
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          double s = cs[q];
          s += __shfl_xor(s, 16); s += __shfl_xor(s, 32);
          const int64_t col = c0 + 16 * VEC * (q / VEC) + VEC * tx + (q % VEC);
          if ((tid & 63) < 16 && col < n) atomicAdd(&dF[col * fdim + d], cd * s);
        }
      }
    }
  }
}

// dF[i][d] += dmu_i w_lin[d]   (mean.linear_mlp: mu = feat . w + b)
template <typename T>
__global__ void grad_feat_mean_kernel(const TaskDesc* tasks, const ModelDev* __restrict__ md, int fdim) {
  const TaskDesc& t = tasks[blockIdx.z];
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)t.n * fdim) return;
  const int64_t i = idx / fdim; const int d = (int)(idx % fdim);
  static_cast<double*>(t.dF)[idx] += static_cast<const double*>(t.dmu)[i] * md->lin_w[d];
}

// d objective / d mu_i.  NLL / EKL: 2 c sum_b e_b alpha_b,i.  EUC: d_i / |d| (0 at d = 0, utils.py safe_l2norm),
// d = mu1 - mu0 = last augmented row; also stores |d| in fnorm[1].
template <typename T>
__global__ __launch_bounds__(256) void dmu_kernel(const TaskDesc* tasks, int obj) {
  __shared__ double sred[4];
  const TaskDesc& t = tasks[blockIdx.x];
  double* dmu = static_cast<double*>(t.dmu);
  if (obj == OBJ_EUC) {
    const T* d = static_cast<const T*>(t.A) + ((int64_t)t.npad + t.naug - 1) * t.ld;
    double q = 0;
    for (int64_t i = threadIdx.x; i < t.n; i += 256) { const double v = (double)d[i]; q += v * v; }
    q = block_sum(q, sred);
    const double nd = sqrt(q);
    if (threadIdx.x == 0) t.fnorm[1] = nd;
    const double inv = nd > 0 ? 1.0 / nd : 0.0;
    for (int64_t i = threadIdx.x; i < t.n; i += 256) dmu[i] = (double)d[i] * inv;
    return;
  }
  const T* al = static_cast<const T*>(t.svec);
  for (int64_t i = threadIdx.x; i < t.n; i += 256) {
    double s = 0;
    for (int b = 0; b < t.naug; ++b) {
      const double e = t.e_all + (b == t.naug - 1 ? t.e_last : 0.0);
      if (e != 0.0) s += e * (double)al[(int64_t)b * t.npad + i];
    }
    dmu[i] = 2.0 * t.coef_c * s;
  }
}

// EUC with an MLP kernel: the kernel part of dF was accumulated with the un-normalised G
__global__ void scale_dF_kernel(const TaskDesc* tasks, int fdim) {
  const TaskDesc& t = tasks[blockIdx.z];
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)t.n * fdim) return;
  const double f = t.fnorm[0];
  static_cast<double*>(t.dF)[idx] *= (f > 0 ? 1.0 / f : 0.0);
}

// MLP backward, one dense+tanh layer:  dz = dout * (1 - out^2) (in place, double)
template <typename T>
__global__ void dense_bwd_dz_kernel(double* __restrict__ dout, const T* __restrict__ out, int64_t count) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= count) return;
  const double o = (double)out[idx];
  dout[idx] *= (1.0 - o * o);
}
// dW[k][o] += sum_i in[i][k] dz[i][o] ; db[o] += sum_i dz[i][o]   (grid.x = k in 0..fin (fin = bias row),
// grid.y = row chunk; threads over o)
template <typename T>
__global__ void dense_bwd_w_kernel(const T* __restrict__ in, const double* __restrict__ dz, int64_t n, int fin,
                                   int fout, double* dW, double* db, int rows_per_block) {
  const int k = blockIdx.x;
  const int64_t i0 = (int64_t)blockIdx.y * rows_per_block;
  int64_t i1 = i0 + rows_per_block; if (i1 > n) i1 = n;
  for (int o = threadIdx.x; o < fout; o += blockDim.x) {
    double s = 0;
    if (k < fin) { for (int64_t i = i0; i < i1; ++i) s += (double)in[i * fin + k] * dz[i * fout + o]; atomicAdd(&dW[(int64_t)k * fout + o], s); }
    else { for (int64_t i = i0; i < i1; ++i) s += dz[i * fout + o]; atomicAdd(&db[o], s); }
  }
}
// din[i][k] = sum_o dz[i][o] w[k][o]
template <typename T>
__global__ void dense_bwd_in_kernel(const double* __restrict__ dz, const T* __restrict__ w, int64_t n, int fin,
                                    int fout, double* din) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n * fin) return;
  const int64_t i = idx / fin; const int k = (int)(idx % fin);
  double s = 0;
  for (int o = 0; o < fout; ++o) s += dz[i * fout + o] * (double)w[(int64_t)k * fout + o];
  din[idx] = s;
}

// per task: reduce tile partials and apply the chain-rule factors; also mean-parameter grads.
// out layout per task (doubles): [lengthscale(n_ls)] [signal_variance] [noise_variance] [constant]
//                                [dot_prod_sigma] [dot_prod_bias] [linear_kernel(fmean)] [linear_bias]
// Column sums of the per-tile partials of a large matrix in two steps: GRAD_PRE workgroups per task each sum every
// GRAD_PRE-th slot (fixed order: deterministic), grad_finalize_kernel then reads GRAD_PRE rows instead of nblk (nblk + 1)
// (one workgroup walking the 4160 slots of a 64-block matrix took 72 us).
constexpr int GRAD_PRE = HBO_GRAD_PRE_ROWS;
__global__ __launch_bounds__(256) void grad_prereduce_kernel(const TaskDesc* tasks, int nacc, const double* partials, int64_t stride_task,
                                                             double* pre) {
  __shared__ double s_part[4][HBO_MAX_FEATURE_DIM + 4];
  const TaskDesc& t = tasks[blockIdx.y];
  const double* part = partials + (int64_t)blockIdx.y * stride_task;
  const int ntile = t.nblk * (t.nblk + 1);
  // thread = (slot lane, column): 256 / 32 = 8 slots in flight per pass over up to 32 columns at a time
  for (int q0 = 0; q0 < nacc; q0 += 32) {
    const int col = q0 + (threadIdx.x & 31), sl = threadIdx.x >> 5;
    double acc = 0;
    if (col < nacc)
      for (int tl = blockIdx.x + GRAD_PRE * sl; tl < ntile; tl += GRAD_PRE * 8) acc += part[(int64_t)tl * nacc + col];
    // sum the 8 slot lanes (two per wave: lanes l and l + 32)
    acc += __shfl_xor(acc, 32);
    if ((threadIdx.x & 63) < 32 && col < nacc) s_part[threadIdx.x >> 6][col] = acc;
    __syncthreads();
    if (threadIdx.x < 32 && col < nacc)
      pre[((int64_t)blockIdx.y * GRAD_PRE + blockIdx.x) * nacc + col] = (s_part[0][col] + s_part[1][col]) + (s_part[2][col] + s_part[3][col]);
    __syncthreads();
  }
}

template <typename T>
__global__ __launch_bounds__(256) void grad_finalize_kernel(const TaskDesc* tasks, const ModelDev* __restrict__ md,
                                                            int fdim, int nacc, int obj, const double* partials,
                                                            int64_t stride_task, double* out, int out_stride,
                                                            double* value_out, int pre_rows) {
  __shared__ double sred[4];
  __shared__ double s_scale;
  const TaskDesc& t = tasks[blockIdx.x];
  const double* part = partials + (int64_t)blockIdx.x * stride_task;
  double* o = out + (int64_t)blockIdx.x * out_stride;
  // two 64-row half-tile slots per lower 128x128 tile -- or the pre_rows rows grad_prereduce_kernel left
  const int ntile = pre_rows > 0 ? pre_rows : t.nblk * (t.nblk + 1);
  const int n_ls = md->n_ls;
  const bool is_dot = (md->kernel_id == HBO_KERNEL_DOT);
  int pos = 0;
  double ls_total = 0;
  // column sums of the per-tile partials [ntile][nacc], QW columns at a time (was: one strided pass and two barriers per
  // column, 55 us at cfg 2)
  __shared__ double s_col[4][32];
  __shared__ double s_tot[HBO_MAX_FEATURE_DIM + 4];   // nacc <= 2 + HBO_MAX_FEATURE_DIM + 1
  constexpr int QW = 32;
  for (int q0 = 0; q0 < nacc; q0 += QW) {
    // a thread sums QW columns of every 256th tile (independent loads), then the columns are reduced over the block
    double sacc[QW];
#pragma unroll
    for (int u = 0; u < QW; ++u) sacc[u] = 0;
    for (int tl = threadIdx.x; tl < ntile; tl += 256) {
      const double* pt = part + (int64_t)tl * nacc + q0;
#pragma unroll
      for (int u = 0; u < QW; ++u) if (q0 + u < nacc) sacc[u] += pt[u];
    }
#pragma unroll
    for (int u = 0; u < QW; ++u) sacc[u] = wave_sum(sacc[u]);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
      for (int u = 0; u < QW; ++u) s_col[threadIdx.x >> 6][u] = sacc[u];
    }
    __syncthreads();
    if (threadIdx.x < QW && q0 + (int)threadIdx.x < nacc)
      s_tot[q0 + threadIdx.x] = (s_col[0][threadIdx.x] + s_col[1][threadIdx.x]) + (s_col[2][threadIdx.x] + s_col[3][threadIdx.x]);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const double f = sqrt(s_tot[nacc - 1]);   // Frobenius slot: EUC scales every kernel-parameter gradient by 1 / |C0 - K1|_F
    s_scale = (obj == OBJ_EUC) ? (f > 0 ? 1.0 / f : 0.0) : 1.0;
    if (obj == OBJ_EUC) { t.fnorm[0] = f; if (value_out) value_out[blockIdx.x] = f + t.fnorm[1]; }
    for (int q = 0; q < nacc - 1; ++q) {
      const double s = s_tot[q] * s_scale;
      if (!is_dot) {
        if (q == 0) o[n_ls] = s / md->sv;                 // signal_variance
        else if (q == 1) o[n_ls + 1] = s;                 // noise_variance
        else {
          const int d = q - 2;
          const double gd = s * (-2.0 * md->inv_ls[d]);   // du/dls_d = -2 ds_d^2 / ls_d
          if (n_ls == 1) ls_total += gd; else o[d] = gd;
        }
      } else {
        if (q == 0) o[n_ls + 3] = s * (-2.0 / (md->dot_sigma * md->dot_sigma * md->dot_sigma));
        else if (q == 1) o[n_ls + 1] = s;
        else o[n_ls + 4] = s * 2.0 * md->dot_bias;
      }
    }
  }
  if (threadIdx.x == 0) {
    if (!is_dot) { if (n_ls == 1) o[0] = ls_total; o[n_ls + 3] = 0; o[n_ls + 4] = 0; }
    else { for (int d = 0; d < n_ls; ++d) o[d] = 0; o[n_ls] = 0; }
  }
  pos = n_ls + 2;
  // mean parameters from d objective / d mu_i (dmu_kernel)
  const double* dmu = static_cast<const double*>(t.dmu);
  double ssum = 0;
  for (int64_t i = threadIdx.x; i < t.n; i += 256) ssum += dmu[i];
  ssum = block_sum(ssum, sred);
  if (threadIdx.x == 0) {
    o[pos] = (md->mean_id == HBO_MEAN_CONSTANT) ? ssum : 0.0;     // constant
  }
  const int lin0 = n_ls + 5;
  const bool lin = (md->mean_id == HBO_MEAN_LINEAR || md->mean_id == HBO_MEAN_LINEAR_MLP);
  const T* fm = static_cast<const T*>(t.Fm);
  for (int d = 0; d < t.fmean; ++d) {
    double s = 0;
    if (lin) for (int64_t i = threadIdx.x; i < t.n; i += 256) s += dmu[i] * (double)fm[i * t.fmean + d];
    s = block_sum(s, sred);
    if (threadIdx.x == 0) o[lin0 + d] = s;
  }
  if (threadIdx.x == 0) o[lin0 + t.fmean] = lin ? ssum : 0.0;
}

// ---------------------------------------------------------------------------------------
// posterior epilogue: mu = Kxq^T alpha + mean(xq); var = kdiag - sum colsq; acquisition.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ double norm_pdf(double x) { return exp(-0.5 * x * x) * 0.3989422804014327; }
__device__ __forceinline__ double norm_cdf(double x) { return 0.5 * erfc(-x * 0.7071067811865476); }

// ---------------------------------------------------------------------------------------
// d acquisition / d (kernel features of the query), one workgroup per query (bayesopt.py:116-125 differentiates
// -ac_func w.r.t. a single x; batches of restarts come as M rows).  With l = W k(X,x), beta = W^T l:
//   mu = k.alpha + m(x), var = k(x,x) - |l|^2, coef_i = a_mu alpha_i - 2 a_var beta_i,
//   SE/Matern: g_d = sum_i coef_i dk/du_i * 2 (f_d - F_id)/ls_d^2;  dot: g = sum_i coef_i F_i/sigma^2 + a_var 2 f/sigma^2.
// ---------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void acq_grad_kernel(AcqGradArgs a, const ModelDev* __restrict__ md) {
  __shared__ double sred[4];
  __shared__ double s_w[256];
  __shared__ double s_fq[HBO_MAX_FEATURE_DIM];
  __shared__ double s_acc[256];
  __shared__ double s_amu, s_avar;
  const int64_t q = blockIdx.x;
  const int tid = threadIdx.x;
  const int fdim = a.fdim;
  const int kid = md->kernel_id;
  const bool is_dot = (kid == HBO_KERNEL_DOT);
  const T* Fq = static_cast<const T*>(a.Fq) + q * fdim;
  const T* F = static_cast<const T*>(a.F);
  const T* Kq = a.Kq ? static_cast<const T*>(a.Kq) + q * a.npad : nullptr;
  const T* L = a.L ? static_cast<const T*>(a.L) + q * a.npad : nullptr;
  const T* B = a.B ? static_cast<const T*>(a.B) + q * a.npad : nullptr;
  const T* al = static_cast<const T*>(a.alpha);
  for (int d = tid; d < fdim; d += 256) s_fq[d] = (double)Fq[d];
  double ka = 0, ll = 0;
  for (int64_t i = tid; i < a.n; i += 256) { ka += (double)Kq[i] * (double)al[i]; const double l = (double)L[i]; ll += l * l; }
  ka = block_sum(ka, sred);
  ll = block_sum(ll, sred);
  if (tid == 0) {
    const double mu = ka + (double)static_cast<const T*>(a.muq)[q];
    const double var = (double)static_cast<const T*>(a.kdiag)[q] - ll;
    const double v2 = (var + a.add_noise) * a.scale;
    const double sd = sqrt(v2);
    double val, amu, asd;
    if (a.acq_id == HBO_ACQ_UCB) { val = mu + a.param * sd; amu = 1.0; asd = a.param; }
    else if (a.acq_id == HBO_ACQ_PI) { val = (mu - a.param) / sd; amu = 1.0 / sd; asd = -(mu - a.param) / (sd * sd); }
    else { const double u = (mu - a.param) / sd; val = sd * (norm_pdf(u) + u * norm_cdf(u)); amu = norm_cdf(u); asd = norm_pdf(u); }
    s_amu = amu; s_avar = asd / (2.0 * sd) * a.scale;
    static_cast<T*>(a.acq_out)[q] = (T)val;
    a.dmu[q] = amu;
  }
  __syncthreads();
  const double amu = s_amu, avar = s_avar;
  // thread layout for the feature reduction: FD = pow2 >= fdim lanes per group, G groups over i
  int FD = 1; while (FD < fdim) FD <<= 1;
  const int G = 256 / FD, grp = tid / FD, dl = tid % FD;
  const double sv = md->sv;
  const double inv_sigma2 = 1.0 / (md->dot_sigma * md->dot_sigma);
  double acc = 0;
  for (int64_t i0 = 0; i0 < a.n; i0 += 256) {
    const int64_t i = i0 + tid;
    double w = 0;
    if (i < a.n) {
      const double coef = amu * (double)al[i] - 2.0 * avar * (double)B[i];
      if (is_dot) w = coef * inv_sigma2;
      else {
        double u = 0;
        for (int d = 0; d < fdim; ++d) { const double df = (s_fq[d] - (double)F[i * fdim + d]) * md->inv_ls[d]; u += df * df; }
        const double k = kfun<double>(kid, u, sv, inv_sigma2, 0.0);
        w = coef * dk_du<double>(kid, u, k, sv) * 2.0;
      }
    }
    __syncthreads();
    s_w[tid] = w;
    __syncthreads();
    const int lim = (int)((a.n - i0) < 256 ? (a.n - i0) : 256);
    if (dl < fdim)
      for (int ii = grp; ii < lim; ii += G) {
        const double fi = (double)F[(i0 + ii) * fdim + dl];
        acc += is_dot ? s_w[ii] * fi : s_w[ii] * (s_fq[dl] - fi);
      }
  }
  __syncthreads();
  s_acc[tid] = acc;
  __syncthreads();
  if (grp == 0 && dl < fdim) {
    double s = 0;
    for (int g = 0; g < G; ++g) s += s_acc[g * FD + dl];
    if (is_dot) s += avar * 2.0 * s_fq[dl] * inv_sigma2;
    else s *= md->inv_ls[dl] * md->inv_ls[dl];
    a.gfeat[q * fdim + dl] = s;
  }
}
// out[q][d] (+)= dmu[q] * lin_w[d]   (linear / linear_mlp mean, mean.py:62-79)
__global__ void acq_grad_mean_kernel(const double* dmu, const ModelDev* __restrict__ md, int64_t M, int fm,
                                     double* out, int accumulate) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= M * fm) return;
  const double v = dmu[idx / fm] * md->lin_w[idx % fm];
  out[idx] = accumulate ? out[idx] + v : v;
}
__global__ void add_inplace_kernel(double* dst, const double* src, int64_t count) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx < count) dst[idx] += src[idx];
}

// mupart[b][q] = sum over the 128 rows i of row block b of Kxq[i][q] * alpha[i]: the posterior mean's product
// Kxq^T alpha (gp.py:300) split by row block, so that its parallelism is (row blocks x queries) -- one thread per
// query walking all n rows takes ~6 ms whatever the number of queries (latency-bound), which an 8192-candidate
// chunk paid in full
template <typename T>
__global__ void post_mupart_kernel(const T* __restrict__ K, int64_t ldq, int n, const T* __restrict__ al, T* mupart, int64_t M) {
  const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int b = blockIdx.y;
  if (q >= M) return;
  const int i0 = b * HBO_TILE, i1 = min(i0 + HBO_TILE, n);
  T s0 = (T)0, s1 = (T)0, s2 = (T)0, s3 = (T)0;
  int i = i0;
  for (; i + 3 < i1; i += 4) {
    s0 += K[(int64_t)i * ldq + q] * al[i];
    s1 += K[(int64_t)(i + 1) * ldq + q] * al[i + 1];
    s2 += K[(int64_t)(i + 2) * ldq + q] * al[i + 2];
    s3 += K[(int64_t)(i + 3) * ldq + q] * al[i + 3];
  }
  for (; i < i1; ++i) s0 += K[(int64_t)i * ldq + q] * al[i];
  mupart[(int64_t)b * ldq + q] = (s0 + s1) + (s2 + s3);
}

template <typename T>
__global__ void post_epilogue_kernel(PostArgs a) {
  const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= a.M) return;
  const T* K = static_cast<const T*>(a.Kxq);
  const T* al = static_cast<const T*>(a.alpha);
  T mu = (T)0;
  if (a.mupart) {   // per-row-block partial sums of Kxq^T alpha (post_mupart_kernel)
    const T* mp = static_cast<const T*>(a.mupart);
    for (int b = 0; b < a.nblk; ++b) mu += mp[(int64_t)b * a.ldq + q];
  } else {
    for (int64_t i = 0; i < a.n; ++i) mu += K[i * a.ldq + q] * al[i];
  }
  mu += static_cast<const T*>(a.muq)[q];
  T var = static_cast<const T*>(a.kdiag)[q];
  const T* cs = static_cast<const T*>(a.colsq);
  T ss = (T)0;
  for (int b = 0; b < a.nblk; ++b) ss += cs[(int64_t)b * a.ldq + q];
  var -= ss;
  if (a.mu_out) static_cast<T*>(a.mu_out)[q] = mu;
  if (a.var_out) static_cast<T*>(a.var_out)[q] = var;
  if (a.acq_out) {
    // GP.predict post-processing (gp.py:607-619) then acfun.py:96-142 in the model dtype
    const T v2 = (var + (T)a.add_noise) * (T)a.scale;
    const T sd = sqrt(v2);
    T r;
    if (a.acq_id == HBO_ACQ_UCB) r = mu + (T)a.param * sd;
    else {
      const T gamma = ((T)a.param - mu) / sd;
      if (a.acq_id == HBO_ACQ_PI) r = -gamma;
      else r = (T)((norm_pdf((double)gamma) - (double)gamma * (1.0 - norm_cdf((double)gamma)))) * sd;
    }
    static_cast<T*>(a.acq_out)[q] = r;
  }
}

// out[a][b] = Kqq[a][b] - sum_i V[i][a] V[i][b]
template <typename T>
__global__ void fullcov_kernel(const T* __restrict__ V, int64_t ldq, int npad, const T* __restrict__ Kqq, int64_t M,
                               T* out) {
  const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t a = blockIdx.y;
  if (b >= M) return;
  T s = (T)0;
  for (int64_t i = 0; i < npad; ++i) s += V[i * ldq + a] * V[i * ldq + b];
  out[a * M + b] = Kqq[a * M + b] - s;
}

template <typename T>
__global__ void extract_lower_kernel(const T* __restrict__ A, int64_t ld, int64_t n, T* out) {
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t r = blockIdx.y;
  if (c >= n) return;
  out[r * n + c] = (c <= r) ? A[r * ld + c] : (T)0;
}
template <typename T>
__global__ void symmetrize_kernel(const T* __restrict__ S, int64_t ld, int64_t n, T* out) {
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t r = blockIdx.y;
  if (c >= n) return;
  // lower tiles of S are valid (full 128x128 tiles on and below the tile diagonal)
  const bool lower_tile = (c / HBO_TILE) <= (r / HBO_TILE);
  out[r * n + c] = lower_tile ? S[r * ld + c] : S[c * ld + r];
}
// dense SPD (n x n host layout, already on device) -> padded A (identity on the padded diagonal)
template <typename T>
__global__ void fill_spd_kernel(const T* __restrict__ a, int64_t n, T* A, int64_t ld, int npad) {
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t r = blockIdx.y;
  if (c >= npad) return;
  T v;
  if (r < n && c < n) v = a[r * n + c];
  else v = (r == c) ? (T)1 : (T)0;
  A[r * ld + c] = v;
}
// augmented rows from b (n x m, row-major): A[(npad+a)*ld + j] = b[j*m + a]; the rest zero
template <typename T>
__global__ void set_aug_kernel(const T* __restrict__ b, int64_t n, int m, T* A, int64_t ld, int npad) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int a = blockIdx.y;
  if (j >= npad) return;
  T v = (T)0;
  if (b && a < m && j < n) v = b[j * m + a];
  A[((int64_t)npad + a) * ld + j] = v;
}
// y = W x (trans=0, rows) or W^T x (trans=1) for lower-triangular W (npad x ld), x: [m][npad]
template <typename T>
__global__ __launch_bounds__(256) void tri_matvec_kernel(const T* __restrict__ W, int64_t ld, int npad,
                                                         const T* __restrict__ x, int64_t xld, int trans,
                                                         T* out, int64_t old) {
  const int col = blockIdx.y;
  if (!trans) {
    // one wave per output row
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= npad) return;
    const int lane = threadIdx.x & 63;
    double s = 0;
    for (int64_t j = lane; j <= r; j += 64) s += (double)W[r * ld + j] * (double)x[(int64_t)col * xld + j];
    s = wave_sum(s);
    if (lane == 0) out[(int64_t)col * old + r] = (T)s;
  } else {
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= npad) return;
    T s = (T)0;
    for (int64_t r = j; r < npad; ++r) s += W[r * ld + j] * x[(int64_t)col * xld + r];
    out[(int64_t)col * old + j] = s;
  }
}

template <typename T, int KID>
void launch_gram_k(const GramArgs& a, const ModelDev* md, dim3 grid, hipStream_t st) {
  if (a.padded || a.tasks) hipLaunchKernelGGL((gram_kernel<T, true, KID>), grid, dim3(256), 0, st, a, md);
  else hipLaunchKernelGGL((gram_kernel<T, false, KID>), grid, dim3(256), 0, st, a, md);
}
template <typename T>
void launch_gram_t(const GramArgs& a, const ModelDev* md, dim3 grid, hipStream_t st) {
  grid.y *= HBO_TILE / GTR;   // callers size the grid in 128x128 tiles; the kernel tiles rows by GTR
  switch (a.kernel_id) {
    case HBO_KERNEL_SE: launch_gram_k<T, HBO_KERNEL_SE>(a, md, grid, st); break;
    case HBO_KERNEL_MATERN32: launch_gram_k<T, HBO_KERNEL_MATERN32>(a, md, grid, st); break;
    case HBO_KERNEL_MATERN52: launch_gram_k<T, HBO_KERNEL_MATERN52>(a, md, grid, st); break;
    default: launch_gram_k<T, HBO_KERNEL_DOT>(a, md, grid, st); break;
  }
}

}  // namespace

#define DISPATCH(dtype, FN, ...) \
  do { if ((dtype) == HBO_F64) FN<double>(__VA_ARGS__); else FN<float>(__VA_ARGS__); } while (0)

void launch_gram(int dtype, const GramArgs& a, const ModelDev* md, dim3 grid, hipStream_t st) {
  DISPATCH(dtype, launch_gram_t, a, md, grid, st);
}
void launch_kdiag(int dtype, const void* f, int64_t n, int fdim, const ModelDev* md, void* out, hipStream_t st) {
  if (n <= 0) return;
  dim3 grid((unsigned)((n + 255) / 256));
  if (dtype == HBO_F64) hipLaunchKernelGGL((kdiag_kernel<double>), grid, dim3(256), 0, st, (const double*)f, n, fdim, md, (double*)out);
  else hipLaunchKernelGGL((kdiag_kernel<float>), grid, dim3(256), 0, st, (const float*)f, n, fdim, md, (float*)out);
}
void launch_dense_tanh(int dtype, const void* in, const void* w, const void* b, void* out, int64_t n, int fin,
                       int fout, hipStream_t st) {
  if (n <= 0) return;
  dim3 grid((unsigned)((n * fout + 255) / 256));
  if (dtype == HBO_F64) hipLaunchKernelGGL((dense_tanh_kernel<double>), grid, dim3(256), 0, st, (const double*)in, (const double*)w, (const double*)b, (double*)out, n, fin, fout);
  else hipLaunchKernelGGL((dense_tanh_kernel<float>), grid, dim3(256), 0, st, (const float*)in, (const float*)w, (const float*)b, (float*)out, n, fin, fout);
}
void launch_mean(int dtype, const void* fm, int64_t n, int fmean, const ModelDev* md, void* mu, hipStream_t st) {
  if (n <= 0) return;
  dim3 grid((unsigned)((n + 255) / 256));
  if (dtype == HBO_F64) hipLaunchKernelGGL((mean_kernel<double>), grid, dim3(256), 0, st, (const double*)fm, n, fmean, md, (double*)mu);
  else hipLaunchKernelGGL((mean_kernel<float>), grid, dim3(256), 0, st, (const float*)fm, n, fmean, md, (float*)mu);
}
void launch_aug_rows(int dtype, const TaskDesc* tasks, int ntasks, int max_npad, const ModelDev* md, hipStream_t st) {
  dim3 grid((max_npad + 255) / 256, 1, ntasks);
  if (dtype == HBO_F64) hipLaunchKernelGGL((aug_rows_kernel<double>), grid, dim3(256), 0, st, tasks, md);
  else hipLaunchKernelGGL((aug_rows_kernel<float>), grid, dim3(256), 0, st, tasks, md);
}
void launch_nll_reduce(int dtype, const TaskDesc* tasks, int ntasks, const int* info, double* out, hipStream_t st) {
  if (dtype == HBO_F64) hipLaunchKernelGGL((nll_reduce_kernel<double>), dim3(ntasks), dim3(256), 0, st, tasks, info, out);
  else hipLaunchKernelGGL((nll_reduce_kernel<float>), dim3(ntasks), dim3(256), 0, st, tasks, info, out);
}
void launch_wt_z(int dtype, const TaskDesc* tasks, int ntasks, int max_nblk, int aug_row, int out_col,
                 int out_ld, hipStream_t st, const void* xover, void* oover) {
  const int max_npad = max_nblk * HBO_TILE;
  dim3 g1(max_nblk, (max_npad + 511) / 512, ntasks);
  dim3 g2((max_npad + 255) / 256, 1, ntasks);
  if (dtype == HBO_F64) {
    hipLaunchKernelGGL((wtz_partial_kernel<double>), g1, dim3(256), 0, st, tasks, aug_row, (const double*)xover);
    hipLaunchKernelGGL((wtz_final_kernel<double>), g2, dim3(256), 0, st, tasks, out_col, out_ld, (double*)oover);
  } else {
    hipLaunchKernelGGL((wtz_partial_kernel<float>), g1, dim3(256), 0, st, tasks, aug_row, (const float*)xover);
    hipLaunchKernelGGL((wtz_final_kernel<float>), g2, dim3(256), 0, st, tasks, out_col, out_ld, (float*)oover);
  }
}
int grad_nacc(int kernel_id, int fdim) { return (kernel_id == HBO_KERNEL_DOT ? 3 : 2 + fdim) + 1; }
namespace {
template <typename T, bool MULTI>
void launch_grad_contract_t(dim3 grid, hipStream_t st, int kernel_id, const TaskDesc* tasks, const ModelDev* md, int fdim,
                            int nacc, int obj, double* partials, int64_t stride_task) {
  switch (kernel_id) {
    case HBO_KERNEL_SE: hipLaunchKernelGGL((grad_contract_kernel<T, MULTI, HBO_KERNEL_SE>), grid, dim3(256), 0, st, tasks, md, fdim, nacc, obj, partials, stride_task); break;
    case HBO_KERNEL_MATERN32: hipLaunchKernelGGL((grad_contract_kernel<T, MULTI, HBO_KERNEL_MATERN32>), grid, dim3(256), 0, st, tasks, md, fdim, nacc, obj, partials, stride_task); break;
    case HBO_KERNEL_MATERN52: hipLaunchKernelGGL((grad_contract_kernel<T, MULTI, HBO_KERNEL_MATERN52>), grid, dim3(256), 0, st, tasks, md, fdim, nacc, obj, partials, stride_task); break;
    default: hipLaunchKernelGGL((grad_contract_kernel<T, MULTI, HBO_KERNEL_DOT>), grid, dim3(256), 0, st, tasks, md, fdim, nacc, obj, partials, stride_task); break;
  }
}
}  // namespace
void launch_grad_contract(int dtype, const TaskDesc* tasks, int ntasks, int max_nblk, const ModelDev* md,
                          int kernel_id, int fdim, int obj, double* partials, int64_t stride_task, hipStream_t st) {
  dim3 grid(2 * max_nblk, max_nblk, ntasks);   // 64-row half tiles: two partial slots per 128x128 tile
  const int nacc = grad_nacc(kernel_id, fdim);
  if (obj == OBJ_NLL) {
    if (dtype == HBO_F64) launch_grad_contract_t<double, false>(grid, st, kernel_id, tasks, md, fdim, nacc, obj, partials, stride_task);
    else launch_grad_contract_t<float, false>(grid, st, kernel_id, tasks, md, fdim, nacc, obj, partials, stride_task);
  } else {
    if (dtype == HBO_F64) launch_grad_contract_t<double, true>(grid, st, kernel_id, tasks, md, fdim, nacc, obj, partials, stride_task);
    else launch_grad_contract_t<float, true>(grid, st, kernel_id, tasks, md, fdim, nacc, obj, partials, stride_task);
  }
}
void launch_dmu(int dtype, const TaskDesc* tasks, int ntasks, int obj, hipStream_t st) {
  if (dtype == HBO_F64) hipLaunchKernelGGL((dmu_kernel<double>), dim3(ntasks), dim3(256), 0, st, tasks, obj);
  else hipLaunchKernelGGL((dmu_kernel<float>), dim3(ntasks), dim3(256), 0, st, tasks, obj);
}
void launch_grad_finalize(int dtype, const TaskDesc* tasks, int ntasks, const ModelDev* md, int kernel_id,
                          int fdim, int obj, const double* partials, int64_t stride_task, double* out,
                          int out_stride, double* value_out, hipStream_t st, double* pre, int max_nblk) {
  const int nacc = grad_nacc(kernel_id, fdim);
  int pre_rows = 0;
  if (pre && max_nblk * (max_nblk + 1) >= 1024) {   // large matrices: column sums in two steps
    hipLaunchKernelGGL(grad_prereduce_kernel, dim3(GRAD_PRE, ntasks), dim3(256), 0, st, tasks, nacc, partials, stride_task, pre);
    partials = pre; stride_task = (int64_t)GRAD_PRE * nacc; pre_rows = GRAD_PRE;
  }
  if (dtype == HBO_F64) hipLaunchKernelGGL((grad_finalize_kernel<double>), dim3(ntasks), dim3(256), 0, st, tasks, md, fdim, nacc, obj, partials, stride_task, out, out_stride, value_out, pre_rows);
  else hipLaunchKernelGGL((grad_finalize_kernel<float>), dim3(ntasks), dim3(256), 0, st, tasks, md, fdim, nacc, obj, partials, stride_task, out, out_stride, value_out, pre_rows);
}
void launch_scale_dF(const TaskDesc* tasks, int ntasks, int64_t max_n, int fdim, hipStream_t st) {
  dim3 grid((unsigned)((max_n * fdim + 255) / 256), 1, ntasks);
  hipLaunchKernelGGL(scale_dF_kernel, grid, dim3(256), 0, st, tasks, fdim);
}
namespace {
template <typename T>
void launch_grad_feat_t(dim3 grid, hipStream_t st, int kernel_id, const TaskDesc* tasks, const ModelDev* md, int fdim, int obj) {
  switch (kernel_id) {
    case HBO_KERNEL_SE: hipLaunchKernelGGL((grad_feat_kernel<T, HBO_KERNEL_SE>), grid, dim3(256), 0, st, tasks, md, fdim, obj); break;
    case HBO_KERNEL_MATERN32: hipLaunchKernelGGL((grad_feat_kernel<T, HBO_KERNEL_MATERN32>), grid, dim3(256), 0, st, tasks, md, fdim, obj); break;
    case HBO_KERNEL_MATERN52: hipLaunchKernelGGL((grad_feat_kernel<T, HBO_KERNEL_MATERN52>), grid, dim3(256), 0, st, tasks, md, fdim, obj); break;
    default: hipLaunchKernelGGL((grad_feat_kernel<T, HBO_KERNEL_DOT>), grid, dim3(256), 0, st, tasks, md, fdim, obj); break;
  }
}
}  // namespace
void launch_grad_feat(int dtype, const TaskDesc* tasks, int ntasks, int max_nblk, const ModelDev* md, int kernel_id,
                      int fdim, int obj, hipStream_t st) {
  dim3 grid(max_nblk, max_nblk, ntasks);
  if (dtype == HBO_F64) launch_grad_feat_t<double>(grid, st, kernel_id, tasks, md, fdim, obj);
  else launch_grad_feat_t<float>(grid, st, kernel_id, tasks, md, fdim, obj);
}
void launch_grad_feat_mean(int dtype, const TaskDesc* tasks, int ntasks, int64_t max_n, const ModelDev* md,
                           int fdim, hipStream_t st) {
  dim3 grid((unsigned)((max_n * fdim + 255) / 256), 1, ntasks);
  if (dtype == HBO_F64) hipLaunchKernelGGL((grad_feat_mean_kernel<double>), grid, dim3(256), 0, st, tasks, md, fdim);
  else hipLaunchKernelGGL((grad_feat_mean_kernel<float>), grid, dim3(256), 0, st, tasks, md, fdim);
}
// one layer of the MLP backward pass for one task; dout (n x fout, double) is turned into dz in place
void launch_dense_bwd(int dtype, const void* in, const void* out, const void* w, double* dout, double* din,
                      double* dW, double* db, int64_t n, int fin, int fout, hipStream_t st) {
  if (n <= 0) return;
  const int64_t cnt = n * fout;
  const int rpb = 256;
  dim3 gw(fin + 1, (unsigned)((n + rpb - 1) / rpb));
  const int thr = fout < 64 ? 64 : (fout > 256 ? 256 : ((fout + 63) / 64) * 64);
  if (dtype == HBO_F64) {
    hipLaunchKernelGGL((dense_bwd_dz_kernel<double>), dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, st, dout, (const double*)out, cnt);
    hipLaunchKernelGGL((dense_bwd_w_kernel<double>), gw, dim3(thr), 0, st, (const double*)in, dout, n, fin, fout, dW, db, rpb);
    if (din) hipLaunchKernelGGL((dense_bwd_in_kernel<double>), dim3((unsigned)((n * fin + 255) / 256)), dim3(256), 0, st, dout, (const double*)w, n, fin, fout, din);
  } else {
    hipLaunchKernelGGL((dense_bwd_dz_kernel<float>), dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, st, dout, (const float*)out, cnt);
    hipLaunchKernelGGL((dense_bwd_w_kernel<float>), gw, dim3(thr), 0, st, (const float*)in, dout, n, fin, fout, dW, db, rpb);
    if (din) hipLaunchKernelGGL((dense_bwd_in_kernel<float>), dim3((unsigned)((n * fin + 255) / 256)), dim3(256), 0, st, dout, (const float*)w, n, fin, fout, din);
  }
}
void launch_acq_grad(int dtype, const AcqGradArgs& a, const ModelDev* md, hipStream_t st) {
  if (a.M <= 0) return;
  if (dtype == HBO_F64) hipLaunchKernelGGL((acq_grad_kernel<double>), dim3((unsigned)a.M), dim3(256), 0, st, a, md);
  else hipLaunchKernelGGL((acq_grad_kernel<float>), dim3((unsigned)a.M), dim3(256), 0, st, a, md);
}
void launch_acq_grad_mean(const double* dmu, const ModelDev* md, int64_t M, int fm, double* out, int accumulate,
                          hipStream_t st) {
  if (M * fm <= 0) return;
  hipLaunchKernelGGL(acq_grad_mean_kernel, dim3((unsigned)((M * fm + 255) / 256)), dim3(256), 0, st, dmu, md, M, fm, out, accumulate);
}
// [sum of the tasks' values, task count, gradient sum in the caller's layout] of one rank's shard, on the device (what the host loop of
// hbo_objective does after the copy back: same order of summation).  One workgroup: the vector has a few dozen entries (plus the
// MLP weights), the task count is at most a few hundred.
__global__ void shard_reduce_kernel(const double* nll, const double* grad, const int* info, int T, int out_stride, const int* map,
                                    const double* mlp, const int* seg, int nseg, double* out, int out_count) {
  const int tid = threadIdx.x;
  for (int i = tid; i < out_count; i += blockDim.x) out[i] = 0.0;
  __syncthreads();
  bool anybad = false;
  for (int k = 0; k < T; ++k) anybad |= info[k] != INT_MAX;
  if (tid == 0) {
    double s = 0.0;
    for (int k = 0; k < T; ++k) s += nll[k];
    out[0] = s; out[1] = (double)T;
  }
  if (grad) {
    for (int j = tid; j < out_stride; j += blockDim.x) {
      const int dst = map[j];
      if (dst < 0) continue;
      double s = 0.0;
      for (int k = 0; k < T; ++k) s += info[k] != INT_MAX ? (double)NAN : grad[(size_t)k * out_stride + j];
      out[2 + dst] += s;
    }
    for (int sgi = 0; sgi < nseg; ++sgi) {
      const int dst = seg[3 * sgi], src = seg[3 * sgi + 1], len = seg[3 * sgi + 2];
      if (dst < 0) continue;
      for (int i = tid; i < len; i += blockDim.x) out[2 + dst + i] = anybad ? (double)NAN : mlp[src + i];
    }
  }
}
void launch_shard_reduce(const double* nll, const double* grad, const int* info, int T, int out_stride, const int* map,
                         const double* mlp, const int* mlp_seg, int n_mlp_seg, double* out, int out_count, hipStream_t st) {
  hipLaunchKernelGGL(shard_reduce_kernel, dim3(1), dim3(256), 0, st, nll, grad, info, T, out_stride, map, mlp, mlp_seg, n_mlp_seg, out, out_count);
}
void launch_add_inplace(double* dst, const double* src, int64_t count, hipStream_t st) {
  if (count <= 0) return;
  hipLaunchKernelGGL(add_inplace_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st, dst, src, count);
}
void launch_post_epilogue(int dtype, const PostArgs& a, hipStream_t st) {
  if (a.M <= 0) return;
  dim3 grid((unsigned)((a.M + 255) / 256));
  if (a.mupart && a.Kxq && a.n > 0) {
    dim3 g2(grid.x, (unsigned)a.nblk);
    if (dtype == HBO_F64) hipLaunchKernelGGL((post_mupart_kernel<double>), g2, dim3(256), 0, st, static_cast<const double*>(a.Kxq), a.ldq, a.n, static_cast<const double*>(a.alpha), static_cast<double*>(a.mupart), a.M);
    else hipLaunchKernelGGL((post_mupart_kernel<float>), g2, dim3(256), 0, st, static_cast<const float*>(a.Kxq), a.ldq, a.n, static_cast<const float*>(a.alpha), static_cast<float*>(a.mupart), a.M);
  }
  if (dtype == HBO_F64) hipLaunchKernelGGL((post_epilogue_kernel<double>), grid, dim3(256), 0, st, a);
  else hipLaunchKernelGGL((post_epilogue_kernel<float>), grid, dim3(256), 0, st, a);
}
void launch_fullcov(int dtype, const void* V, int64_t ldq, int npad, const void* Kqq, int64_t M, void* out,
                    hipStream_t st) {
  if (M <= 0) return;
  dim3 grid((unsigned)((M + 255) / 256), (unsigned)M);
  if (dtype == HBO_F64) hipLaunchKernelGGL((fullcov_kernel<double>), grid, dim3(256), 0, st, (const double*)V, ldq, npad, (const double*)Kqq, M, (double*)out);
  else hipLaunchKernelGGL((fullcov_kernel<float>), grid, dim3(256), 0, st, (const float*)V, ldq, npad, (const float*)Kqq, M, (float*)out);
}
void launch_extract_lower(int dtype, const void* A, int64_t ld, int64_t n, void* out, hipStream_t st) {
  if (n <= 0) return;
  dim3 grid((unsigned)((n + 255) / 256), (unsigned)n);
  if (dtype == HBO_F64) hipLaunchKernelGGL((extract_lower_kernel<double>), grid, dim3(256), 0, st, (const double*)A, ld, n, (double*)out);
  else hipLaunchKernelGGL((extract_lower_kernel<float>), grid, dim3(256), 0, st, (const float*)A, ld, n, (float*)out);
}
void launch_symmetrize_from_lower(int dtype, const void* S, int64_t ld, int64_t n, void* out, hipStream_t st) {
  if (n <= 0) return;
  dim3 grid((unsigned)((n + 255) / 256), (unsigned)n);
  if (dtype == HBO_F64) hipLaunchKernelGGL((symmetrize_kernel<double>), grid, dim3(256), 0, st, (const double*)S, ld, n, (double*)out);
  else hipLaunchKernelGGL((symmetrize_kernel<float>), grid, dim3(256), 0, st, (const float*)S, ld, n, (float*)out);
}
void launch_fill_spd(int dtype, const void* a, int64_t n, void* A, int64_t ld, int npad, hipStream_t st) {
  dim3 grid((npad + 255) / 256, npad);
  if (dtype == HBO_F64) hipLaunchKernelGGL((fill_spd_kernel<double>), grid, dim3(256), 0, st, (const double*)a, n, (double*)A, ld, npad);
  else hipLaunchKernelGGL((fill_spd_kernel<float>), grid, dim3(256), 0, st, (const float*)a, n, (float*)A, ld, npad);
}
void launch_set_aug(int dtype, const void* b, int64_t n, int m, void* A, int64_t ld, int npad, hipStream_t st) {
  dim3 grid((npad + 255) / 256, HBO_TILE);
  if (dtype == HBO_F64) hipLaunchKernelGGL((set_aug_kernel<double>), grid, dim3(256), 0, st, (const double*)b, n, m, (double*)A, ld, npad);
  else hipLaunchKernelGGL((set_aug_kernel<float>), grid, dim3(256), 0, st, (const float*)b, n, m, (float*)A, ld, npad);
}
void launch_tri_matvec(int dtype, const void* W, int64_t ld, int npad, const void* x, int64_t xld, int m,
                       int trans, void* out, int64_t old, hipStream_t st) {
  dim3 grid(trans ? (npad + 255) / 256 : (npad + 3) / 4, m);
  if (dtype == HBO_F64) hipLaunchKernelGGL((tri_matvec_kernel<double>), grid, dim3(256), 0, st, (const double*)W, ld, npad, (const double*)x, xld, trans, (double*)out, old);
  else hipLaunchKernelGGL((tri_matvec_kernel<float>), grid, dim3(256), 0, st, (const float*)W, ld, npad, (const float*)x, xld, trans, (float*)out, old);
}
