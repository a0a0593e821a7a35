// Posterior and acquisition kernels on gfx950: mu = Kxq^T alpha + mean(xq), var = k(xq, xq) - column sums of V^2, fused
// EI / PI / UCB, the full-covariance epilogue, and d acquisition / d x_query.
//
// Reference restated: hyperbo/gp_utils/gp.py:242-305 (predict), bo_utils/acfun.py:96-142 (acquisition),
// bo_utils/bayesopt.py:116-125 (the gradient L-BFGS-B takes of -ac_func).
#include "kernfun.h"

namespace {
// ---------------------------------------------------------------------------------------
// posterior epilogue: mu = Kxq^T alpha + mean(xq); var = kdiag - sum colsq; acquisition.
// ---------------------------------------------------------------------------------------

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
  const ExpCoef ec = hbo_exp_coef();
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
        const double k = kfun(kid, u, sv, inv_sigma2, 0.0, ec);
        w = coef * dk_du(kid, u, k, sv, ec) * 2.0;
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


// out[col] = W x[col] (trans = 0) or W^T x[col] (trans = 1), W lower triangular, one right-hand side per blockIdx.y: the row
// append (one vector).
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

// The same products for MC right-hand sides per pass over W (the acquisition gradient asks for l = W k(X, x_q) and beta = W^T l for
// every restart point x_q; one pass per point re-read all of W: N = 8000, 16 points: 4.3 ms).
//   forward (W x):  a workgroup owns 16 rows, four per wave; lanes stride over j; 4 x MC accumulators per lane; fp64 sums
//   transposed (W^T x): a workgroup owns 64 columns (lane = column: the row reads of W coalesce), the right-hand sides of 256 rows at a time
//   are staged in LDS (read by every lane), each wave takes 64 of those rows eight at a time; the waves' partial sums meet in LDS
template <typename T, int MC>
__global__ __launch_bounds__(256) void tri_matmat_fwd_kernel(const T* __restrict__ W, int64_t ld, int npad, const T* __restrict__ x, int64_t xld,
                                                             int m, T* out, int64_t old) {
  const int c0 = blockIdx.y * MC;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t r0 = (int64_t)blockIdx.x * 16 + wave * 4;
  if (r0 >= npad) return;
  double s[4][MC];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int c = 0; c < MC; ++c) s[a][c] = 0;
  const int64_t rmax = r0 + 3;
  for (int64_t j = lane; j <= rmax; j += 64) {
    double xv[MC];
#pragma unroll
    for (int c = 0; c < MC; ++c) xv[c] = c0 + c < m ? (double)x[(int64_t)(c0 + c) * xld + j] : 0.0;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const double w = j <= r0 + a ? (double)W[(r0 + a) * ld + j] : 0.0;
#pragma unroll
      for (int c = 0; c < MC; ++c) s[a][c] += w * xv[c];
    }
  }
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int c = 0; c < MC; ++c) {
      const double v = wave_sum(s[a][c]);
      if (lane == 0 && c0 + c < m) out[(int64_t)(c0 + c) * old + r0 + a] = (T)v;
    }
}
template <typename T, int MC>
__global__ __launch_bounds__(256) void tri_matmat_trans_kernel(const T* __restrict__ W, int64_t ld, int npad, const T* __restrict__ x, int64_t xld,
                                                               int m, T* out, int64_t old) {
  __shared__ double red[4][MC][64];
  __shared__ double xs[MC][256];          // the right-hand sides of the current 256 rows (read by every lane: LDS broadcast)
  const int c0 = blockIdx.y * MC;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t j0 = (int64_t)blockIdx.x * 64, j = j0 + lane;
  double s[MC];
#pragma unroll
  for (int c = 0; c < MC; ++c) s[c] = 0;
  for (int64_t rc = j0 / 256 * 256; rc < npad; rc += 256) {     // npad is a multiple of 128: the last chunk may be half
    __syncthreads();
#pragma unroll
    for (int c = 0; c < MC; ++c) {
      const int64_t r = rc + threadIdx.x;
      xs[c][threadIdx.x] = (c0 + c < m && r < npad) ? (double)x[(int64_t)(c0 + c) * xld + r] : 0.0;
    }
    __syncthreads();
    // each wave takes 64 rows of the chunk, eight at a time: their loads of W are in flight together
#pragma unroll 2
    for (int g = 0; g < 8; ++g) {
      const int rl = wave * 64 + g * 8;
      T w[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) { const int64_t r = rc + rl + u; w[u] = (r < npad && r >= j) ? W[r * ld + j] : (T)0; }
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int c = 0; c < MC; ++c) s[c] += (double)w[u] * xs[c][rl + u];
    }
  }
#pragma unroll
  for (int c = 0; c < MC; ++c) red[wave][c][lane] = s[c];
  __syncthreads();
  if (wave == 0)
#pragma unroll
    for (int c = 0; c < MC; ++c)
      if (c0 + c < m) out[(int64_t)(c0 + c) * old + j] = (T)(((red[0][c][lane] + red[1][c][lane]) + red[2][c][lane]) + red[3][c][lane]);
}

}  // namespace
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
void launch_add_inplace(double* dst, const double* src, int64_t count, hipStream_t st) {
  if (count <= 0) return;
  hipLaunchKernelGGL(add_inplace_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st, dst, src, count);
}
namespace {
// split-K posterior product (few candidate columns): the chunks' partial products are summed, squared and summed over the 128
// rows of a row block -- the same [nblk][ldq] partials the one-pass product leaves for the epilogue
template <typename T>
__global__ __launch_bounds__(256) void post_colsq_split_kernel(const T* __restrict__ vpart, int npad, int64_t ldq, int kchunk, T* __restrict__ colsq) {
  __shared__ double red[4][64];
  const int i = blockIdx.y, col = blockIdx.x * 64 + (threadIdx.x & 63), rg = threadIdx.x >> 6;
  const int nch = (i + kchunk) / kchunk;
  double s = 0;
  for (int r = rg * 32; r < rg * 32 + 32; ++r) {
    const int64_t off = ((int64_t)i * HBO_TILE + r) * ldq + col;
    double v = 0;
    for (int ch = 0; ch < nch; ++ch) v += (double)vpart[(int64_t)ch * npad * ldq + off];
    s += v * v;
  }
  red[rg][threadIdx.x & 63] = s;
  __syncthreads();
  if (rg == 0) colsq[(int64_t)i * ldq + col] = (T)(red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
}
}  // namespace
void launch_post_colsq_split(int dtype, const void* vpart, int npad, int64_t ldq, int mpad, int nblk, int kchunk, void* colsq, hipStream_t st) {
  const dim3 grid(mpad / 64, nblk);
  if (dtype == HBO_F64) hipLaunchKernelGGL(post_colsq_split_kernel<double>, grid, dim3(256), 0, st, (const double*)vpart, npad, ldq, kchunk, (double*)colsq);
  else hipLaunchKernelGGL(post_colsq_split_kernel<float>, grid, dim3(256), 0, st, (const float*)vpart, npad, ldq, kchunk, (float*)colsq);
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
void launch_tri_matvec(int dtype, const void* W, int64_t ld, int npad, const void* x, int64_t xld, int m,
                       int trans, void* out, int64_t old, hipStream_t st) {
  if (m <= 0) return;
  if (m == 1) {
    dim3 grid(trans ? (npad + 255) / 256 : (npad + 3) / 4, 1);
    if (dtype == HBO_F64) hipLaunchKernelGGL((tri_matvec_kernel<double>), grid, dim3(256), 0, st, (const double*)W, ld, npad, (const double*)x, xld, trans, (double*)out, old);
    else hipLaunchKernelGGL((tri_matvec_kernel<float>), grid, dim3(256), 0, st, (const float*)W, ld, npad, (const float*)x, xld, trans, (float*)out, old);
    return;
  }
  constexpr int MC = 8;
  const dim3 grid(trans ? npad / 64 : (npad + 15) / 16, (m + MC - 1) / MC);
  if (dtype == HBO_F64) {
    if (trans) hipLaunchKernelGGL((tri_matmat_trans_kernel<double, MC>), grid, dim3(256), 0, st, (const double*)W, ld, npad, (const double*)x, xld, m, (double*)out, old);
    else hipLaunchKernelGGL((tri_matmat_fwd_kernel<double, MC>), grid, dim3(256), 0, st, (const double*)W, ld, npad, (const double*)x, xld, m, (double*)out, old);
  } else {
    if (trans) hipLaunchKernelGGL((tri_matmat_trans_kernel<float, MC>), grid, dim3(256), 0, st, (const float*)W, ld, npad, (const float*)x, xld, m, (float*)out, old);
    else hipLaunchKernelGGL((tri_matmat_fwd_kernel<float, MC>), grid, dim3(256), 0, st, (const float*)W, ld, npad, (const float*)x, xld, m, (float*)out, old);
  }
}
