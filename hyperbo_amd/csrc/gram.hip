// HBM-bound kernels of the Gram stage on gfx950: pairwise-kernel Gram build (LDS-tiled X blocks, coalesced 16-byte stores,
// fused +(sigma^2+eps) I), MLP features, mean / augmented residual rows, NLL reduction, s = W^T z, and the dense helpers of
// hbo_spd_solve / hbo_cache_export.
//
// Reference restated: hyperbo/gp_utils/kernel.py:29-145 (Gram), basis_functions.py:24-36 (MLP), mean.py:30-79,
// basics/linalg.py:36-69 (jitter), objectives.py:144-156 (NLL).
#include "kernfun.h"

namespace {

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
    md += (int64_t)blockIdx.z * g.model_stride;
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
    auto step = [&](int dd) {
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
    };
    // (not unrolled: by 4 or 16 the LDS reads of all the steps are hoisted -- 232-238 VGPRs, two waves per SIMD)
    for (int dd = 0; dd < dlim; ++dd) step(dd);
  }

  const ExpCoef ec = hbo_exp_coef();
  const T sv = (T)md->sv;
  const T inv_sigma2 = (T)(1.0 / (md->dot_sigma * md->dot_sigma));
  const T bias2 = (T)(md->dot_bias * md->dot_bias);
  const T diag_add = (T)(md->noise + md->eps);
  // Interior tiles -- all 64 x 128 elements are data and none is on the diagonal -- take a straight epilogue: the per-element
  // "inside the data? on the diagonal?" 64-bit compares, selects and exec-mask branches of the general one were a third of the
  // kernel's instructions (it is bound by VALU issue: ~3700 instructions per wave x 4 cycles x 16 waves per SIMD = the 100 us
  // of the N = 8192 build), and all but ~3 % of the tiles of a large matrix are interior.
  const bool interior = PADDED && r0 + GTR <= n1 && c0 + HBO_TILE <= n2 && !(g.symmetric && r0 < c0 + HBO_TILE && c0 < r0 + GTR);
  if (interior) {
    // (a scheduling barrier per row of 8 elements: left alone the scheduler interleaves all 32 exponentials: 232 VGPRs)
#pragma unroll
    for (int a = 0; a < GRA; ++a) {
      __builtin_amdgcn_sched_barrier(0);
      T* orow = out + (r0 + ty + 16 * a) * ldo + c0 + VEC * tx;
#pragma unroll
      for (int qb = 0; qb < 8 / VEC; ++qb) {
        vec_t vv;
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
#ifdef HBO_GRAM_NOEXP
          vv[e] = acc[a][qb * VEC + e] * sv;
#else
          vv[e] = kfun(kid, acc[a][qb * VEC + e], sv, inv_sigma2, bias2, ec);
#endif
        }
#ifdef HBO_GRAM_NOSTORE
        if (vv[0] == (T)123.456) gst(reinterpret_cast<vec_t*>(orow + 16 * VEC * qb), vv);
#else
        gst(reinterpret_cast<vec_t*>(orow + 16 * VEC * qb), vv);
#endif
      }
    }
    return;
  }
#pragma unroll
  for (int a = 0; a < GRA; ++a) {
    __builtin_amdgcn_sched_barrier(0);
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
          v = kfun(kid, acc[a][qb * VEC + e], sv, inv_sigma2, bias2, ec);
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
__global__ void mean_kernel(const T* __restrict__ fm, int64_t n, int fmean, const ModelDev* __restrict__ md,
                            T* out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  out[i] = mean_at<T>(md, fm, fmean, i);
}

// augmented tile-row: row b < naug holds  aug_src[b*n + j] + e_b * mu_j  (see TaskDesc); everything else zero.
template <typename T>
__global__ void aug_rows_kernel(const TaskDesc* tasks, const ModelDev* __restrict__ md, int model_stride) {
  const TaskDesc& t = tasks[blockIdx.z];
  md += (int64_t)blockIdx.z * model_stride;
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
      v = ys[(int64_t)(a == t.naug - 1 ? t.last_src : a) * t.n + j] + e * mu;
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

// rows of n elements -> rows of npad elements, zero padded (the data rows of a divergence objective with more than 127 aligned columns)
template <typename T>
__global__ void expand_rows_kernel(const T* __restrict__ src, int64_t n, int npad, T* __restrict__ dst) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= npad) return;
  dst[(int64_t)blockIdx.y * npad + j] = j < n ? src[(int64_t)blockIdx.y * n + j] : (T)0;
}
// value[0] += coef * sum over `count` rows of npad elements of z^2 (one workgroup: the rows are few and short)
template <typename T>
__global__ __launch_bounds__(256) void add_sumsq_kernel(const T* __restrict__ z, int64_t total, double coef, double* value) {
  __shared__ double sred[4];
  double q = 0;
  for (int64_t i = threadIdx.x; i < total; i += 256) { const double v = (double)z[i]; q += v * v; }
  q = block_sum(q, sred);
  if (threadIdx.x == 0) value[0] += coef * q;
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
void launch_expand_rows(int dtype, const void* src, int64_t n, int npad, void* dst, int count, hipStream_t st) {
  if (count <= 0) return;
  const dim3 grid((npad + 255) / 256, count);
  if (dtype == HBO_F64) hipLaunchKernelGGL((expand_rows_kernel<double>), grid, dim3(256), 0, st, (const double*)src, n, npad, (double*)dst);
  else hipLaunchKernelGGL((expand_rows_kernel<float>), grid, dim3(256), 0, st, (const float*)src, n, npad, (float*)dst);
}
void launch_add_sumsq(int dtype, const void* z, int npad, int count, double coef, double* value, hipStream_t st) {
  if (count <= 0) return;
  if (dtype == HBO_F64) hipLaunchKernelGGL((add_sumsq_kernel<double>), dim3(1), dim3(256), 0, st, (const double*)z, (int64_t)npad * count, coef, value);
  else hipLaunchKernelGGL((add_sumsq_kernel<float>), dim3(1), dim3(256), 0, st, (const float*)z, (int64_t)npad * count, coef, value);
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
// hbo_tune "poison" (tests): everything an evaluation is about to recompute is set to NaN first -- the lower triangles of A (Gram -> L)
// and W (L^-1; the zeros above its diagonal are an invariant of the buffer and stay), all of S (scratch -> K^-1), alpha and d f / d mu.
// A launch that silently skips work (a tile counter shared by two launches, a K range cut short) then shows up as NaN instead of hiding
// behind the previous evaluation's identical numbers in the same buffers.  The augmented tile-row (rows npad...) is left alone: its
// unused rows are independent of every result, and the fp32 split's scale measurement reads the whole tile-row.
template <typename T>
__global__ __launch_bounds__(256) void poison_kernel(const TaskDesc* tasks) {
  const TaskDesc& t = tasks[blockIdx.z];
  const int64_t row = blockIdx.x;
  if (row >= t.npad) return;
  const T nanv = (T)NAN;
  T* A = static_cast<T*>(t.A) + row * t.ld;
  T* W = t.W ? static_cast<T*>(t.W) + row * t.ld : nullptr;
  T* S = t.S ? static_cast<T*>(t.S) + row * t.ld : nullptr;
  for (int64_t cc = threadIdx.x; cc < t.npad; cc += blockDim.x) {
    if (cc <= row) { A[cc] = nanv; if (W) W[cc] = nanv; }
    if (S) S[cc] = nanv;
  }
  if (row == 0) {
    const int ncol = t.nvec ? t.nvec : (t.naug > 0 ? t.naug : 1);
    if (t.svec) for (int64_t cc = threadIdx.x; cc < (int64_t)t.npad * ncol; cc += blockDim.x) static_cast<T*>(t.svec)[cc] = nanv;
    if (t.dmu) for (int64_t cc = threadIdx.x; cc < t.npad; cc += blockDim.x) static_cast<double*>(t.dmu)[cc] = (double)NAN;
  }
}
void launch_poison(int dtype, const TaskDesc* tasks, int ntasks, int max_npad, hipStream_t st) {
  dim3 grid(max_npad, 1, ntasks);
  if (dtype == HBO_F64) hipLaunchKernelGGL((poison_kernel<double>), grid, dim3(256), 0, st, tasks);
  else hipLaunchKernelGGL((poison_kernel<float>), grid, dim3(256), 0, st, tasks);
}
void launch_aug_rows(int dtype, const TaskDesc* tasks, int ntasks, int max_npad, const ModelDev* md, hipStream_t st, int model_stride) {
  dim3 grid((max_npad + 255) / 256, 1, ntasks);
  if (dtype == HBO_F64) hipLaunchKernelGGL((aug_rows_kernel<double>), grid, dim3(256), 0, st, tasks, md, model_stride);
  else hipLaunchKernelGGL((aug_rows_kernel<float>), grid, dim3(256), 0, st, tasks, md, model_stride);
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