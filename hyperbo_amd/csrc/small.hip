// One workgroup per sub-dataset: the whole NLL (+ gradient) evaluation of a task with n <= 128 points in ONE launch for the batch.
//
// The reference's real pre-training regime is many small tasks -- 10-64 sub-datasets of 50-100 points per step
// (hyperbo/basics/data_utils.py:72-100 sub-samples every task down to `batch_size`; gp_test.py:58-148) -- where the blocked
// pipeline of objective.hip is 13 launches of 4-25 us each.  Here every stage of
//   objectives.py:144-156 (NLL of one sub-dataset) and its jax.value_and_grad (gp.py:134)
// runs inside one 512-thread workgroup with the matrix resident in LDS as packed 16x16 tiles:
//   Gram + (noise + eps) I  ->  potf2 (the leaf chain of chol.hip, panel_dev.h)  ->  W = L^-1 in place (block columns, last to
//   first)  ->  z = W r, s = W^T z  ->  K^-1 = W^T W in place  ->  sum_ij G_ij dK_ij/dtheta with G = lh K^-1 - c s s^T  ->
//   the task's gradient block in the layout of grad_finalize_kernel + its NLL + its info word.
// Same leaf algebra as the blocked path (kfun / dk_du of kernfun.h, leaf_cholesky4), same outputs in the same buffers (d_nll,
// d_gradout, d_info): everything downstream -- the host sum over tasks, the sharded reduce -- is unchanged.  Sums are taken in a
// different order than the tiled kernels take them: results agree to rounding (tests: 1e-12 of the blocked path), not to the bit.
#include "kernfun.h"
#include <type_traits>

namespace {
#include "panel_dev.h"

constexpr int SMALL_THREADS = 512;
constexpr int SMALL_WAVES = SMALL_THREADS / 64;
constexpr int SM_RED = HBO_MAX_FEATURE_DIM + 8;

template <typename T>
constexpr int small_lds_bytes() {
  return (36 + 8 + 7) * TILE_ELEMS * (int)sizeof(T)   // matrix tiles, leaf inverses, one scratch block column
         + 4 * NB * (int)sizeof(T)                     // 1 / diag L, r, z, s
         + DC * SXS * (int)sizeof(T)                   // staged feature chunk
         + (SMALL_WAVES * (DC + 4) + SM_RED) * (int)sizeof(double);
}

// (I, J) of the lower 16 x 16 tiles in tri_index order, as arithmetic on a compile-time index.  (As constexpr tables the compiler turned
// `thalf ? I[2 e + 1] : I[2 e]` into a LOAD of I[2 e + thalf] from constant memory: two dependent loads with a wait per element and
// phase -- 18 of them in a row in the Gram phase, again in the contraction -- where two immediates and a select were meant.)
#define HBO_TRI_I(k) ((k) < 1 ? 0 : (k) < 3 ? 1 : (k) < 6 ? 2 : (k) < 10 ? 3 : (k) < 15 ? 4 : (k) < 21 ? 5 : (k) < 28 ? 6 : 7)
#define HBO_TRI_J(k) ((k) - HBO_TRI_I(k) * (HBO_TRI_I(k) + 1) / 2)
struct SmallArgs {
  const TaskDesc* tasks;
  const ModelDev* md;
  int* info;           // [T] first failing pivot + 1 (INT_MAX: fine)
  double* nll_out;     // [T]
  double* grad_out;    // [T][out_stride] or null (value only)
  int out_stride;
  int fdim;
  int write_back;      // MLP models: K^-1 (full square) -> S, s -> svec, d f / d mu -> dmu for the feature-gradient kernels that follow
};

template <typename T, int KID>
__global__ __launch_bounds__(SMALL_THREADS) void small_eval_kernel(SmallArgs g) {
  typedef typename Mma<T>::acc_t acc_t;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  T* sT = reinterpret_cast<T*>(smem);            // 36 lower tiles [16][17]: K -> L -> W -> K^-1
  T* sMall = sT + 36 * TILE_ELEMS;               // leaf inverses M_j = L_jj^-1
  T* sCol = sMall + 8 * TILE_ELEMS;              // scratch block column of the in-place inverse
  T* sDinv = sCol + 7 * TILE_ELEMS;              // 1 / L_ii
  T* sR = sDinv + NB;                            // r = sum_a y_a + e mu
  T* sZ = sR + NB;                               // z = L^-1 r
  T* sS = sZ + NB;                               // s = K^-1 r
  T* sX = sS + NB;                               // [DC][SXS] staged (scaled) features of all 128 rows
  double* swred = reinterpret_cast<double*>(sX + DC * SXS);   // [waves][DC + 4]
  double* s_tot = swred + SMALL_WAVES * (DC + 4);             // [SM_RED]

  const TaskDesc& t = g.tasks[blockIdx.x];
  const ModelDev* md = g.md;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, lq = lane >> 4;
  const int n = t.n, fdim = g.fdim;
  constexpr int kid = KID;
  constexpr bool is_dot = (kid == HBO_KERNEL_DOT);
  const T* F = static_cast<const T*>(t.F);
  const int nleaf = (n + 15) / 16;
  const int ntile = nleaf * (nleaf + 1) / 2;     // lower tiles that hold data, in tri_index order
  int* info_slot = g.info + blockIdx.x;

  // Element phases (Gram, contraction): the lower tiles are walked in tri_index order, thread = one element position of a 16 x 16
  // tile, two tiles per pass of the 512 threads -- element e of the thread: tile (tid >> 8) + 2 e, row 16 I + ri, column 16 J + ci.
  // (A 4 x 8 micro-tile over the 128 x 128 square, as gram_kernel has it, spends half of its lanes on tiles above the diagonal.)
  constexpr int EPT = 18;                        // 36 tiles / 2
  const int ri = (tid & 255) >> 4, ci = tid & 15, thalf = tid >> 8;
  auto tile_ij = [](int tix, int& I, int& J) {
    int ii = 0;
    while ((ii + 1) * (ii + 2) / 2 <= tix) ++ii;
    I = ii; J = tix - ii * (ii + 1) / 2;
  };

  auto stage = [&](int d0, bool scale) {         // features [d0, d0 + DC) of every row into sX[dd][row]
    const int dd = tid & 15, rr0 = tid >> 4;
    const int d = d0 + dd;
    const T sc = (scale && d < fdim) ? (T)md->inv_ls[d] : (T)1;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int row = rr0 + 32 * q;
      T v = (T)0;
      if (row < n && d < fdim) v = gld(F + (int64_t)row * fdim + d) * sc;
      sX[dd * SXS + row] = v;
    }
  };
  auto tile_at = [&](int row, int col) -> T* {   // element (row, col), row tile >= column tile
    return sT + tri_index(row >> 4, col >> 4) * TILE_ELEMS + (row & 15) * TS + (col & 15);
  };

  const ExpLit ec;
  const T sv = (T)md->sv;
  const T inv_sigma2 = (T)(1.0 / (md->dot_sigma * md->dot_sigma));
  const T bias2 = (T)(md->dot_bias * md->dot_bias);

  // ---- residual row (aug_rows_kernel) and the Gram matrix + (noise + eps) I with identity padding (gram_kernel) ----------------
  if (tid == 0) *info_slot = 0x7fffffff;         // (the caller no longer clears the info words of a batch that comes here)
  if (tid < NB) {
    T v = (T)0;
    if (tid < n) {
      const T e = (T)(t.e_all + t.e_last);       // one augmented row (NLL): b = naug - 1
      v = gld(static_cast<const T*>(t.ysum) + tid) + e * mean_at<T>(md, static_cast<const T*>(t.Fm), t.fmean, tid);
    }
    sR[tid] = v;
  }
  // scaled squared distances (dot product: inner products) and covariances of the thread's elements: kept in registers for the
  // contraction at the end (the blocked pipeline recomputes both from the features)
  T uu[EPT], kk[EPT];
#pragma unroll
  for (int e = 0; e < EPT; ++e) uu[e] = (T)0;
  // row / column of element e: tile thalf + 2 e -- e is a compile-time index in the unrolled loops and thalf is uniform per wave,
  // so the tile's (I, J) are two scalar selects from the tables (kept as index arrays they cost 36 registers for the whole kernel)
#define EROW(e) (16 * (thalf ? HBO_TRI_I(2 * (e) + 1) : HBO_TRI_I(2 * (e))) + ri)
#define ECOL(e) (16 * (thalf ? HBO_TRI_J(2 * (e) + 1) : HBO_TRI_J(2 * (e))) + ci)
  for (int d0 = 0; d0 < fdim; d0 += DC) {
    __syncthreads();
    stage(d0, !is_dot);
    __syncthreads();
    const int dlim = (fdim - d0) < DC ? (fdim - d0) : DC;
    for (int dd = 0; dd < dlim; ++dd) {
      // (all 18 elements, also those of tiles beyond the data: straight-line code; their values are never used)
      T xa[8], xb[EPT];
#pragma unroll
      for (int q = 0; q < 8; ++q) xa[q] = sX[dd * SXS + 16 * q + ri];
#pragma unroll
      for (int e = 0; e < EPT; ++e) xb[e] = sX[dd * SXS + ECOL(e)];
#pragma unroll
      for (int e = 0; e < EPT; ++e) {
        const T a = thalf ? xa[HBO_TRI_I(2 * e + 1)] : xa[HBO_TRI_I(2 * e)], b = xb[e];
        if (is_dot) uu[e] += a * b;
        else { const T df = a - b; uu[e] += df * df; }
      }
    }
  }
  {
    const T diag_add = (T)(md->noise + md->eps);
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
      const int tix = thalf + 2 * e;
      kk[e] = (T)0;
      if (tix >= ntile) continue;
      const int row = EROW(e), col = ECOL(e);
      T v;
      if (row < n && col < n) {
        kk[e] = kfun(kid, uu[e], sv, inv_sigma2, bias2, ec);
        v = kk[e];
        if (row == col) v += diag_add;
      } else {
        v = (row == col) ? (T)1 : (T)0;
      }
      sT[tix * TILE_ELEMS + ri * TS + ci] = v;   // (diagonal tiles: both triangles)
    }
  }
  __syncthreads();

#if defined(HBO_SMALL_STOP) && HBO_SMALL_STOP == 1   // (phase timing builds: tools/README.md)
  return;
#endif
  // ---- potf2 in LDS: the loop of chol.hip:potf2_body on tiles that are already resident, all leaf inverses kept.  The helper waves
  //      also build W = L^-1 row block by row block IN PLACE behind the leaf chain:  W[i,j] = -M_i sum_{k=j..i-1} L[i,k] W[k,j]  needs
  //      row block i of L (final since step j of each of its tiles), the rows < i of W and the leaf inverse M_i -- all there one
  //      step after leaf i; nothing else reads row block i of L any more, so W[i,j] overwrites L[i,j] (the waves take different j:
  //      the L[i,k] a wave reads have k >= its j).  (A separate in-LDS inverse after the loop took 8 of the kernel's 56 us.)
  T* Wb = static_cast<T*>(t.W);                  // (leaf_cholesky4 also stores each leaf inverse to W: 8 x 256 elements, unused here)
  const int64_t ldw = t.ld;
  auto factor_leaf = [&](int jb) {               // wave 0 only
    T* dt = sT + tri_index(jb, jb) * TILE_ELEMS;
    acc_t acc;
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = dt[Mma<T>::crow(lane, r) * TS + l15];
    const int bad = leaf_cholesky4<T>(acc, dt, sMall + jb * TILE_ELEMS, sDinv + jb * 16, Wb + (int64_t)(jb * 16) * ldw + jb * 16, ldw, lane);
    if (bad >= 0 && lane == 0) atomicMin(info_slot, jb * 16 + bad + 1);
  };
  auto solve_tile = [&](int jb, int R) {         // rows of tile (R, jb): X = A M_jb^T, in place
    T* xt = sT + tri_index(R, jb) * TILE_ELEMS;
    const T* sM = sMall + jb * TILE_ELEMS;
    acc_t acc = {0, 0, 0, 0};
#pragma unroll
    for (int kk4 = 0; kk4 < 4; ++kk4) acc = Mma<T>::mma(xt[l15 * TS + kk4 * 4 + lq], sM[l15 * TS + kk4 * 4 + lq], acc);
#pragma unroll
    for (int r = 0; r < 4; ++r) xt[Mma<T>::crow(lane, r) * TS + l15] = acc[r];
  };
  auto update_tile = [&](int jb, int I, int J) { // C[I][J] -= X_I X_J^T (K = 16)
    T* ct = sT + tri_index(I, J) * TILE_ELEMS;
    const T* at = sT + tri_index(I, jb) * TILE_ELEMS;
    const T* bt = sT + tri_index(J, jb) * TILE_ELEMS;
    acc_t acc;
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = ct[Mma<T>::crow(lane, r) * TS + l15];
#pragma unroll
    for (int kk4 = 0; kk4 < 4; ++kk4) acc = Mma<T>::mma(-at[l15 * TS + kk4 * 4 + lq], bt[l15 * TS + kk4 * 4 + lq], acc);
#pragma unroll
    for (int r = 0; r < 4; ++r) ct[Mma<T>::crow(lane, r) * TS + l15] = acc[r];
  };
  // S_j = sum_{k=j..i-1} L[i,k] W[k,j]  (W[k,k] = M_k) into the scratch column, one tile per call
  auto wrow_sum = [&](int i, int j) {
    acc_t acc = {0, 0, 0, 0};
    for (int k = j; k < i; ++k) {
      const T* lt = sT + tri_index(i, k) * TILE_ELEMS;
      const T* wt = (k == j) ? sMall + j * TILE_ELEMS : sT + tri_index(k, j) * TILE_ELEMS;
#pragma unroll
      for (int kk4 = 0; kk4 < 4; ++kk4) acc = Mma<T>::mma(lt[l15 * TS + kk4 * 4 + lq], wt[(kk4 * 4 + lq) * TS + l15], acc);
    }
    T* st = sCol + j * TILE_ELEMS;
#pragma unroll
    for (int r = 0; r < 4; ++r) st[Mma<T>::crow(lane, r) * TS + l15] = acc[r];
  };
  // W[i,j] = -M_i S_j, over L[i,j]
  auto wrow_finish = [&](int i, int j) {
    const T* sM = sMall + i * TILE_ELEMS;
    const T* st = sCol + j * TILE_ELEMS;
    acc_t acc = {0, 0, 0, 0};
#pragma unroll
    for (int kk4 = 0; kk4 < 4; ++kk4) acc = Mma<T>::mma(-sM[l15 * TS + kk4 * 4 + lq], st[(kk4 * 4 + lq) * TS + l15], acc);
    T* ot = sT + tri_index(i, j) * TILE_ELEMS;
#pragma unroll
    for (int r = 0; r < 4; ++r) ot[Mma<T>::crow(lane, r) * TS + l15] = acc[r];
  };
  // helper waves 1, 2, 3, 5, 6, 7 (wave 4 shares wave 0's SIMD: the leaf chain runs alone there, as in potf2_body)
  const bool helper = wave != 0 && wave != 4;
  const int hw = wave < 4 ? wave - 1 : wave - 2;  // 0..5
  if (wave == 0 && nleaf > 0) factor_leaf(0);
  __syncthreads();
  for (int jb = 0; jb < nleaf; ++jb) {
    for (int R = jb + 1 + wave; R < nleaf; R += SMALL_WAVES) solve_tile(jb, R);
    __syncthreads();
    // (C) trailing update; wave 0 takes the next diagonal tile and factors it; the helpers then the sums of row block jb of W
    //     (leaf jb and everything left of it are final)
    const int m = nleaf - 1 - jb;
    const int ntl = m * (m + 1) / 2;             // tile 0 is (jb+1, jb+1): wave 0's, followed by the next leaf
    if (wave == 0) {
      if (m > 0) { update_tile(jb, jb + 1, jb + 1); factor_leaf(jb + 1); }
    } else if (helper) {
      for (int tix = 1 + hw; tix < ntl; tix += 6) {
        int ii = 0;
        while ((ii + 1) * (ii + 2) / 2 <= tix) ++ii;
        update_tile(jb, jb + 1 + ii, jb + 1 + tix - ii * (ii + 1) / 2);
      }
      for (int j = hw; j < jb; j += 6) wrow_sum(jb, j);
    }
    __syncthreads();
    if (helper) for (int j = hw; j < jb; j += 6) wrow_finish(jb, j);
    // (the next iteration's solve_tile touches column jb + 1 only: rows > jb + 1; row block jb is not read again)
  }
  __syncthreads();

#if defined(HBO_SMALL_STOP) && HBO_SMALL_STOP == 2
  return;
#endif
  // ---- log-determinant from 1 / diag L; the diagonal tiles take their leaf inverses: sT now holds W = L^-1 (lower tiles) -----------
  double ld_sum = 0;
  if (tid < n) ld_sum = -log((double)sDinv[tid]);
  for (int j = wave; j < nleaf; j += SMALL_WAVES) {
    T* dt = sT + tri_index(j, j) * TILE_ELEMS;
    const T* sM = sMall + j * TILE_ELEMS;
    for (int e = lane; e < 256; e += 64) dt[(e >> 4) * TS + (e & 15)] = sM[(e >> 4) * TS + (e & 15)];
  }
  __syncthreads();

  // ---- z = W r, s = W^T z (sixteen lanes per row / column, four rows per wave and pass), the quadratic form and the NLL ------------
  const int np = nleaf * 16;
  for (int i0 = 0; i0 < np; i0 += 32) {
    const int i = i0 + (tid >> 4);
    T a = (T)0;
    if (i < np) for (int j = l15; j <= i; j += 16) a += *tile_at(i, j) * sR[j];
    a += __shfl_xor(a, 1); a += __shfl_xor(a, 2); a += __shfl_xor(a, 4); a += __shfl_xor(a, 8);
    if (l15 == 0 && i < np) sZ[i] = a;
  }
  if (tid >= np && tid < NB) { sZ[tid] = (T)0; sS[tid] = (T)0; }
  __syncthreads();
  for (int j0 = 0; j0 < np; j0 += 32) {
    const int j = j0 + (tid >> 4);
    T a = (T)0;
    if (j < np) for (int i = j + l15; i < np; i += 16) a += *tile_at(i, j) * sZ[i];
    a += __shfl_xor(a, 1); a += __shfl_xor(a, 2); a += __shfl_xor(a, 4); a += __shfl_xor(a, 8);
    if (l15 == 0 && j < np) sS[j] = a;
  }
  double qpart = 0;
  if (tid < NB) { const double z = tid < n ? (double)sZ[tid] : 0.0; qpart = z * z; }
  {
    const double q = wave_sum(qpart), l = wave_sum(ld_sum);
    if (lane == 0) { swred[wave * (DC + 4)] = q; swred[wave * (DC + 4) + 1] = l; }
  }
  __syncthreads();
  if (tid == 0) {
    double q = 0, l = 0;
    for (int w = 0; w < SMALL_WAVES; ++w) { q += swred[w * (DC + 4)]; l += swred[w * (DC + 4) + 1]; }
    double v = t.coef_c * q + 2.0 * t.coef_lh * l + t.coef_const;
    if (*info_slot != 0x7fffffff) v = NAN;
    g.nll_out[blockIdx.x] = v;
  }
  if (!g.grad_out) return;

#if defined(HBO_SMALL_STOP) && HBO_SMALL_STOP == 4
  return;
#endif
  // ---- K^-1 = W^T W on the lower tiles, in place: every wave holds its tiles in registers until all reads are done ---------------
  {
    constexpr int MAXT = (36 + SMALL_WAVES - 1) / SMALL_WAVES;
    acc_t out[MAXT];
#pragma unroll
    for (int s = 0; s < MAXT; ++s) {
      out[s] = (acc_t){0, 0, 0, 0};
      const int tix = wave + SMALL_WAVES * s;
      if (tix < ntile) {
        int i, j;
        tile_ij(tix, i, j);
        for (int k = i; k < nleaf; ++k) {
          const T* wi = sT + tri_index(k, i) * TILE_ELEMS;
          const T* wj = sT + tri_index(k, j) * TILE_ELEMS;
#pragma unroll
          for (int kk4 = 0; kk4 < 4; ++kk4) out[s] = Mma<T>::mma(wi[(kk4 * 4 + lq) * TS + l15], wj[(kk4 * 4 + lq) * TS + l15], out[s]);
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < MAXT; ++s) {
      const int tix = wave + SMALL_WAVES * s;
      if (tix < ntile) {
        T* ot = sT + tix * TILE_ELEMS;
#pragma unroll
        for (int r = 0; r < 4; ++r) ot[Mma<T>::crow(lane, r) * TS + l15] = out[s][r];
      }
    }
  }
  __syncthreads();
  if (g.write_back) {
    T* S = static_cast<T*>(t.S);
    for (int e0 = tid; e0 < NB * NB; e0 += SMALL_THREADS) {
      const int row = e0 >> 7, col = e0 & 127;
      if (row < n && col < n) gst(S + (int64_t)row * t.ld + col, row >= col ? *tile_at(row, col) : *tile_at(col, row));
    }
    if (tid < NB) {
      gst(static_cast<T*>(t.svec) + tid, sS[tid]);
      if (tid < n) gst(static_cast<double*>(t.dmu) + tid, 2.0 * t.coef_c * (t.e_all + t.e_last) * (double)sS[tid]);
    }
  }

#if defined(HBO_SMALL_STOP) && HBO_SMALL_STOP == 5
  return;
#endif
  // ---- contraction sum_ij G_ij dK_ij / dtheta over the lower triangle, G = lh K^-1 - c s s^T (grad_contract_kernel), from the
  //      distances and covariances of the Gram phase ----------------------------------------------------------------------------
  const T lh = (T)t.coef_lh, cc = (T)t.coef_c;
  double a_gk = 0, a_tr = 0, a_g = 0;
#pragma unroll
  for (int e = 0; e < EPT; ++e) {
    const int tix = thalf + 2 * e;
    T gw = (T)0;
    if (tix < ntile) {
      const int row = EROW(e), col = ECOL(e);
      if (row < n && col <= row) {
        const T u = uu[e], k = kk[e];
        const T G0 = lh * sT[tix * TILE_ELEMS + ri * TS + ci] - cc * (sS[row] * sS[col]);
        const T G = (row == col) ? G0 : G0 * (T)2;      // an element below the diagonal stands for its mirror image too
        if (is_dot) { a_gk += (double)(G * u); a_g += (double)G; }
        else { a_gk += (double)(G * k); gw = G * dk_du(kid, u, k, sv, ec); }
        if (row == col) a_tr += (double)G;
      }
    }
    uu[e] = gw;                                  // (re-uses the register)
  }
  a_gk = wave_sum(a_gk); a_tr = wave_sum(a_tr); a_g = wave_sum(a_g);
  __syncthreads();                               // (swred held the NLL partial sums: thread 0 has read them)
  if (lane == 0) { swred[wave * (DC + 4)] = a_gk; swred[wave * (DC + 4) + 1] = a_tr; swred[wave * (DC + 4) + 2] = a_g; }
  __syncthreads();
  if (tid < 3) {
    double s = 0;
    for (int w = 0; w < SMALL_WAVES; ++w) s += swred[w * (DC + 4) + tid];
    s_tot[tid == 2 ? (is_dot ? 2 : SM_RED - 1) : tid] = s;      // slots as in grad_contract_kernel: [0] G.K, [1] tr G, dot [2] sum G
  }
  if (!is_dot) {
    for (int d0 = 0; d0 < fdim; d0 += DC) {
      // (up to DC features: the scaled chunk the Gram phase staged is still in sX)
      if (fdim > DC) { __syncthreads(); stage(d0, true); __syncthreads(); }
      const int dlim = (fdim - d0) < DC ? (fdim - d0) : DC;
      for (int dd = 0; dd < dlim; ++dd) {
        T s = (T)0;
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
          if (thalf + 2 * e >= ntile) continue;
          const T df = sX[dd * SXS + EROW(e)] - sX[dd * SXS + ECOL(e)];
          s += uu[e] * df * df;
        }
        const double ws = wave_sum((double)s);
        if (lane == 0) swred[wave * (DC + 4) + 4 + dd] = ws;
      }
      __syncthreads();
      if (tid < dlim) {
        double s = 0;
        for (int w = 0; w < SMALL_WAVES; ++w) s += swred[w * (DC + 4) + 4 + tid];
        s_tot[2 + d0 + tid] = s;
      }
    }
  }
  __syncthreads();

  // ---- the task's gradient block (grad_finalize_kernel: chain-rule factors, mean parameters from d f / d mu = 2 c e s) -----------
  double* o = g.grad_out + (int64_t)blockIdx.x * g.out_stride;
  const int n_ls = md->n_ls;
  if (tid == 0) {
    if (!is_dot) {
      o[n_ls] = s_tot[0] / md->sv;
      o[n_ls + 1] = s_tot[1];
      double ls_total = 0;
      for (int d = 0; d < fdim; ++d) {
        const double gd = s_tot[2 + d] * (-2.0 * md->inv_ls[d]);
        if (n_ls == 1) ls_total += gd; else o[d] = gd;
      }
      if (n_ls == 1) o[0] = ls_total;
      o[n_ls + 3] = 0; o[n_ls + 4] = 0;
    } else {
      o[n_ls + 3] = s_tot[0] * (-2.0 / (md->dot_sigma * md->dot_sigma * md->dot_sigma));
      o[n_ls + 1] = s_tot[1];
      o[n_ls + 4] = s_tot[2] * 2.0 * md->dot_bias;
      for (int d = 0; d < n_ls; ++d) o[d] = 0;
      o[n_ls] = 0;
    }
  }
  // mean parameters: wave w takes output w - 1 (-1: the sum of d f / d mu = constant / bias; d >= 0: linear weight d)
  const double dmu_scale = 2.0 * t.coef_c * (t.e_all + t.e_last);
  const bool lin = (md->mean_id == HBO_MEAN_LINEAR || md->mean_id == HBO_MEAN_LINEAR_MLP);
  const T* fm = static_cast<const T*>(t.Fm);
  const int lin0 = n_ls + 5;
  for (int d = wave - 1; d < (lin ? t.fmean : 0); d += SMALL_WAVES) {
    double s = 0;
    for (int i = lane; i < n; i += 64) {
      double v = dmu_scale * (double)sS[i];
      if (d >= 0) v *= (double)gld(fm + (int64_t)i * t.fmean + d);
      s += v;
    }
    s = wave_sum(s);
    if (lane == 0) {
      if (d < 0) { o[n_ls + 2] = (md->mean_id == HBO_MEAN_CONSTANT) ? s : 0.0; o[lin0 + t.fmean] = lin ? s : 0.0; }
      else o[lin0 + d] = s;
    }
  }
  if (!lin && tid == 0) for (int d = 0; d < t.fmean; ++d) o[lin0 + d] = 0.0;
}

#undef EROW
#undef ECOL

template <typename T>
void small_eval_t(const SmallArgs& a, int ntasks, int kernel_id, hipStream_t st) {
  static unsigned long long seen = 0;
  if (hbo_first_use_on_device(seen)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&small_eval_kernel<T, HBO_KERNEL_SE>), hipFuncAttributeMaxDynamicSharedMemorySize, small_lds_bytes<T>());
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&small_eval_kernel<T, HBO_KERNEL_MATERN32>), hipFuncAttributeMaxDynamicSharedMemorySize, small_lds_bytes<T>());
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&small_eval_kernel<T, HBO_KERNEL_MATERN52>), hipFuncAttributeMaxDynamicSharedMemorySize, small_lds_bytes<T>());
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&small_eval_kernel<T, HBO_KERNEL_DOT>), hipFuncAttributeMaxDynamicSharedMemorySize, small_lds_bytes<T>());
  }
  const dim3 grid(ntasks), block(SMALL_THREADS);
  const int lds = small_lds_bytes<T>();
  switch (kernel_id) {
    case HBO_KERNEL_SE: hipLaunchKernelGGL((small_eval_kernel<T, HBO_KERNEL_SE>), grid, block, lds, st, a); break;
    case HBO_KERNEL_MATERN32: hipLaunchKernelGGL((small_eval_kernel<T, HBO_KERNEL_MATERN32>), grid, block, lds, st, a); break;
    case HBO_KERNEL_MATERN52: hipLaunchKernelGGL((small_eval_kernel<T, HBO_KERNEL_MATERN52>), grid, block, lds, st, a); break;
    default: hipLaunchKernelGGL((small_eval_kernel<T, HBO_KERNEL_DOT>), grid, block, lds, st, a); break;
  }
}
}  // namespace

// dynamic LDS one workgroup of small_eval_kernel asks for (fp64 132 KB, fp32 68 KB: gfx950's 160 KB holds both; the callers compare it
// with the device's per-workgroup limit and send the batch through the blocked pipeline where it does not fit)
int small_eval_lds(int dtype) { return dtype == HBO_F64 ? small_lds_bytes<double>() : small_lds_bytes<float>(); }
// NLL (+ gradient block) of every task of a batch whose tasks all have n <= 128, one workgroup per task, one launch.
void launch_small_eval(int dtype, const TaskDesc* tasks, int ntasks, const ModelDev* md, int kernel_id, int fdim, int* info,
                       double* nll_out, double* grad_out, int out_stride, int write_back, hipStream_t st) {
  SmallArgs a = {tasks, md, info, nll_out, grad_out, out_stride, fdim, write_back};
  if (dtype == HBO_F64) small_eval_t<double>(a, ntasks, kernel_id, st);
  else small_eval_t<float>(a, ntasks, kernel_id, st);
}
