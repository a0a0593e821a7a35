// Gradient kernels of the GP objectives on gfx950: the contraction sum_ij G_ij dK_ij / dtheta over the lower tiles (K recomputed
// on the fly, K^-1 read once), d / d features for the MLP kernels with the dense-layer backward pass, d / d mean parameters,
// the two-stage reductions, and the device-side sum over a rank's tasks for the sharded objective.
//
// Reference restated: jax.value_and_grad of hyperbo/gp_utils/objectives.py:109-210 (gp.py:134), with the VJP of
// basics/linalg.py:129-171 and the zero-distance rule of linalg.py:173-197.
#include "kernfun.h"

namespace {
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
    if (t.nvec) { stride = t.npad; count = t.nvec - 1; return static_cast<const T*>(t.svec) + t.npad; }   // data rows in svec columns 1..m
    stride = t.ld; count = t.naug - 1;
    return static_cast<const T*>(t.A) + (int64_t)t.npad * t.ld;
  }
  stride = t.npad; count = t.nvec ? t.nvec : t.naug;
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
  // (literal coefficients and no straight path for interior tiles, unlike gram_kernel: with either the kernel goes from 150 to
  //  180-232 VGPRs and from three to two waves per SIMD, and it is bound by latency, not by instruction count -- 0.255 ms at cfg 2
  //  against 0.28-0.29 with one or both; profiles/r04_elementwise.md)
  const ExpLit ec;
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
        const T k = kfun(kid, u, sv, inv_sigma2, bias2, ec);
        const T G0 = euc ? (k + (row == col ? noise : (T)0) - outer[q]) : (lh * kinv_row[q] - cc * outer[q]);
        const T G = G0 * wt;
        if (MULTI) a_fro += (double)(G0 * G);
        if (is_dot) { a_gk += (double)(G * u); a_g += (double)G; }
        else { a_gk += (double)(G * k); gw = G * dk_du(kid, u, k, sv, ec); }
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
  const ExpLit ec;
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
        const T k = kfun(kid, u, sv, inv_sigma2, bias2, ec);
        T outer = (T)0;
        for (int b = 0; b < nvec; ++b) outer += sv_[(int64_t)b * vstride + row] * sv_[(int64_t)b * vstride + col];
        const T G = euc ? (k + (row == col ? noise : (T)0) - outer) : (lh * S[row * t.ld + col] - cc * outer);
        if (is_dot) g = G;
        else g = G * dk_du(kid, u, k, sv, ec);
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
  // column sums of the per-tile partials [ntile][nacc]: thread = (column of a 32-wide strip, one of 8 row lanes); a row lane walks every
  // 8th row with independent, coalesced loads, the lanes meet in LDS in a fixed order (deterministic).  (Before: a thread summed 32
  // columns of every 256th row -- 32 loads with a wait behind each -- and every column went through a wave reduction: 12-25 us at the
  // tail of every evaluation of the blocked pipeline.)
  __shared__ double s_col[8][32];
  __shared__ double s_tot[HBO_MAX_FEATURE_DIM + 4];   // nacc <= 2 + HBO_MAX_FEATURE_DIM + 1
  constexpr int QW = 32;
  for (int q0 = 0; q0 < nacc; q0 += QW) {
    const int col = q0 + (threadIdx.x & 31), rl = threadIdx.x >> 5;
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    if (col < nacc) {
      int tl = rl;
      for (; tl + 24 < ntile; tl += 32) {
        a0 += part[(int64_t)tl * nacc + col]; a1 += part[(int64_t)(tl + 8) * nacc + col];
        a2 += part[(int64_t)(tl + 16) * nacc + col]; a3 += part[(int64_t)(tl + 24) * nacc + col];
      }
      for (; tl < ntile; tl += 8) a0 += part[(int64_t)tl * nacc + col];
    }
    s_col[rl][threadIdx.x & 31] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (threadIdx.x < QW && q0 + (int)threadIdx.x < nacc) {
      double tsum = 0;
#pragma unroll
      for (int r8 = 0; r8 < 8; ++r8) tsum += s_col[r8][threadIdx.x];
      s_tot[q0 + threadIdx.x] = tsum;
    }
    __syncthreads();
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
  {   // (four loads in flight per thread: one workgroup walks all n entries, 32 dependent round trips at n = 8192 otherwise)
    const int64_t nn = t.n;
    double s1 = 0, s2 = 0, s3 = 0;
    int64_t i = threadIdx.x;
    for (; i + 768 < nn; i += 1024) { ssum += dmu[i]; s1 += dmu[i + 256]; s2 += dmu[i + 512]; s3 += dmu[i + 768]; }
    for (; i < nn; i += 256) ssum += dmu[i];
    ssum = (ssum + s1) + (s2 + s3);
  }
  ssum = block_sum(ssum, sred);
  if (threadIdx.x == 0) {
    o[pos] = (md->mean_id == HBO_MEAN_CONSTANT) ? ssum : 0.0;     // constant
  }
  const int lin0 = n_ls + 5;
  const bool lin = (md->mean_id == HBO_MEAN_LINEAR || md->mean_id == HBO_MEAN_LINEAR_MLP);
  const T* fm = static_cast<const T*>(t.Fm);
  for (int d = 0; d < t.fmean; ++d) {
    double s = 0;
    if (lin) {
      const int64_t nn = t.n; const int fmn = t.fmean;
      double s1 = 0, s2 = 0, s3 = 0;
      int64_t i = threadIdx.x;
      for (; i + 768 < nn; i += 1024) {
        s += dmu[i] * (double)fm[i * fmn + d]; s1 += dmu[i + 256] * (double)fm[(i + 256) * fmn + d];
        s2 += dmu[i + 512] * (double)fm[(i + 512) * fmn + d]; s3 += dmu[i + 768] * (double)fm[(i + 768) * fmn + d];
      }
      for (; i < nn; i += 256) s += dmu[i] * (double)fm[i * fmn + d];
      s = (s + s1) + (s2 + s3);
    }
    s = block_sum(s, sred);
    if (threadIdx.x == 0) o[lin0 + d] = s;
  }
  if (threadIdx.x == 0) o[lin0 + t.fmean] = lin ? ssum : 0.0;
}

}  // namespace
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