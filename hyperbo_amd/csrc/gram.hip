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

// ---- fp32 Gram of a stationary covariance on the matrix cores (round 6) ----------------------------------------------------------
// With 32 or more features the distance loop of gram_kernel -- two VALU instructions per pair and feature -- is what the fp32 Gram
// costs (cfg 3: 64 tanh features, cross Gram 0.58 ms per chunk of 8192 candidates = 0.93 TB/s written).  Here the pair term comes off
// the bf16 matrix cores: u_ij = |a_i|^2 + |b_j|^2 - 2 a_i . b_j with a = x / lengthscale in fp32, the norms summed in fp32 by the
// thread that stages the row, and the dot product from the EXACT three-way bf16 split of both operands (hbo_split3: six
// v_mfma_f32_32x32x16_bf16 per product, fp32 accumulate -- every bit of both fp32 factors, as post3.hip).  128 x 128 tile per
// workgroup, 4 waves of 2 x 2 MFMA blocks, features staged 32 at a time.  What changes against the direct form sum (a - b)^2 is the
// ROUNDING of u for close pairs (absolute error ~ 1e-7 |a|^2 instead of relative 1e-7): measured on cfg 3's features max |dK| 7.5e-6
// against fp64 where the direct fp32 form has 1.2e-7 -- inside the 2e-5 the fp32 Gram is held to, which is why fp64 and narrow
// feature spaces keep gram_kernel.  The diagonal of a symmetric Gram gets u = 0 exactly.  Reference: kernel.py:63-123.
typedef __bf16 gm_bf16x8 __attribute__((ext_vector_type(8)));
typedef float gm_f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned short gm_u16x8 __attribute__((ext_vector_type(8)));
constexpr int GM_KC = 16;     // features per staged chunk (one MFMA k step): 38 KB of LDS, FOUR workgroups per CU -- with 32 (62 KB, two per CU) the
                              // waves spent half of their 41 000 cycles waiting (SQ_WAIT_ANY / SQ_WAVE_CYCLES, profiles/r06_gram_mfma.md)
constexpr int GM_ROWB = 2 * GM_KC + 16;   // bytes of one LDS row of one plane: the chunk's bf16 + 16 bytes (the b128 fragment reads of a lane group cover all 64 banks once)
constexpr int GM_LPR = GM_KC / 4;         // lanes per row of the staging (one float4 each)
constexpr int GM_RPP = 256 / GM_LPR;      // rows per staging pass
constexpr int GM_NQ = 256 / GM_RPP;       // passes: the tile's 128 rows, then its 128 columns
template <int KID>
__global__ __launch_bounds__(256, 3) void gram_mfma_kernel(GramArgs g, const ModelDev* __restrict__ md) {
  __shared__ __attribute__((aligned(16))) unsigned char sP[2 * 3 * 128 * GM_ROWB];   // [operand][plane][row]
  __shared__ float sN[2][128];
  const int ti = blockIdx.y;
  const int tj = g.symmetric ? (int)((blockIdx.x + blockIdx.y) % gridDim.x) : (int)blockIdx.x;
  const float* x1; const float* x2; float* out; int64_t n1, n2, ldo; int64_t e1, e2;
  if (g.tasks) {
    const TaskDesc& t = g.tasks[blockIdx.z];
    md += (int64_t)blockIdx.z * g.model_stride;
    if (ti >= t.nblk || tj >= t.nblk) return;
    x1 = x2 = static_cast<const float*>(t.F);
    out = static_cast<float*>(t.A);
    n1 = n2 = t.n; ldo = t.ld; e1 = e2 = t.npad;
  } else {
    x1 = static_cast<const float*>(g.x1); x2 = static_cast<const float*>(g.x2); out = static_cast<float*>(g.out);
    n1 = g.n1; n2 = g.n2; ldo = g.ldo; e1 = g.padded ? g.n1pad : g.n1; e2 = g.padded ? g.n2pad : g.n2;
  }
  const int64_t r0 = (int64_t)ti * HBO_TILE, c0 = (int64_t)tj * HBO_TILE;
  if (g.symmetric && tj > ti) return;   // entirely above the diagonal
  if (r0 >= e1 || c0 >= e2) return;
  const int fdim = g.fdim;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, l32 = lane & 31, lh = lane >> 5;

  // staging: GM_LPR consecutive lanes read the chunk's features of one row (a float4 each), a pass of the 256 threads covers GM_RPP rows;
  // the first half of the passes are the tile's rows (x1), the second half its columns (x2).  Every thread keeps the partial |row|^2 of its
  // pieces; GM_LPR-lane sums at the end.
  const int sc4 = tid % GM_LPR, srow = tid / GM_LPR;
  const bool vec_ok = (fdim & 3) == 0 && ((reinterpret_cast<unsigned long long>(x1) | reinterpret_cast<unsigned long long>(x2)) & 15) == 0;
  float nrm[GM_NQ];
#pragma unroll
  for (int q = 0; q < GM_NQ; ++q) nrm[q] = 0.f;
  auto load_chunk = [&](int d0, float4 (&v)[GM_NQ]) {
#pragma unroll
    for (int q = 0; q < GM_NQ; ++q) {
      const int opq = q / (GM_NQ / 2), rr = srow + GM_RPP * (q % (GM_NQ / 2));
      const float* src = opq ? x2 : x1;
      const int64_t nrow = opq ? n2 : n1;
      const int64_t gr = (opq ? c0 : r0) + rr;
      const float* xr = src + (gr < nrow ? gr : (nrow > 0 ? nrow - 1 : 0)) * (int64_t)fdim;   // (clamped: always a valid address)
      const int d = d0 + 4 * sc4;
      if (vec_ok && d + 4 <= fdim) v[q] = *reinterpret_cast<const float4*>(xr + d);
      else { v[q].x = d < fdim ? xr[d] : 0.f; v[q].y = d + 1 < fdim ? xr[d + 1] : 0.f; v[q].z = d + 2 < fdim ? xr[d + 2] : 0.f; v[q].w = d + 3 < fdim ? xr[d + 3] : 0.f; }
      if (gr >= nrow) v[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  gm_f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[a][b][q] = 0.f;

  float4 vcur[GM_NQ];
  load_chunk(0, vcur);
  for (int d0 = 0; d0 < fdim; d0 += GM_KC) {
    const int d = d0 + 4 * sc4;
    float isc[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) isc[e] = (d + e < fdim) ? (float)md->inv_ls[d + e] : 0.f;
    __syncthreads();   // the previous chunk's fragments are read
#pragma unroll
    for (int q = 0; q < GM_NQ; ++q) {
      const float sx[4] = {vcur[q].x * isc[0], vcur[q].y * isc[1], vcur[q].z * isc[2], vcur[q].w * isc[3]};
      unsigned short ph[4], pm[4], pl[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) { nrm[q] = fmaf(sx[e], sx[e], nrm[q]); hbo_split3(sx[e], ph[e], pm[e], pl[e]); }
      const int opq = q / (GM_NQ / 2), rr = srow + GM_RPP * (q % (GM_NQ / 2));
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        const unsigned short* w = p == 0 ? ph : (p == 1 ? pm : pl);
        uint2 pk;
        pk.x = (unsigned)w[0] | ((unsigned)w[1] << 16); pk.y = (unsigned)w[2] | ((unsigned)w[3] << 16);
        *reinterpret_cast<uint2*>(sP + (size_t)((opq * 3 + p) * 128 + rr) * GM_ROWB + 8 * sc4) = pk;
      }
    }
    if (d0 + GM_KC < fdim) load_chunk(d0 + GM_KC, vcur);   // in flight beside this chunk's products
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < GM_KC / 16; ++ks) {
      // the COLUMN block is the MFMA's row operand: a lane then holds 4 consecutive Gram columns of one Gram row (16-byte stores)
      gm_bf16x8 fa[3][2], fb[3][2];
#pragma unroll
      for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          fa[p][t] = *reinterpret_cast<const gm_bf16x8*>(sP + (size_t)((1 * 3 + p) * 128 + wn * 64 + t * 32 + l32) * GM_ROWB + ks * 32 + lh * 16);
          fb[p][t] = *reinterpret_cast<const gm_bf16x8*>(sP + (size_t)((0 * 3 + p) * 128 + wm * 64 + t * 32 + l32) * GM_ROWB + ks * 32 + lh * 16);
        }
      constexpr int PA[6] = {2, 0, 1, 1, 0, 0};   // smallest products first: l h', h l', m m', m h', h m', h h'
      constexpr int PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
      for (int q = 0; q < 6; ++q)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < 2; ++b)
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[PA[q]][a], fb[PB[q]][b], acc[a][b], 0, 0, 0);
    }
  }
#pragma unroll
  for (int q = 0; q < GM_NQ; ++q) {
    float s_ = nrm[q];
#pragma unroll
    for (int o = 1; o < GM_LPR; o *= 2) s_ += __shfl_xor(s_, o);
    if (sc4 == 0) sN[q / (GM_NQ / 2)][srow + GM_RPP * (q % (GM_NQ / 2))] = s_;
  }
  __syncthreads();

  const float sv = (float)md->sv;
  const float diag_add = (float)(md->noise + md->eps);
  const bool on_diag = g.symmetric && ti == tj;
  const bool interior = r0 + HBO_TILE <= n1 && c0 + HBO_TILE <= n2 && !on_diag && (ldo & 3) == 0 && (reinterpret_cast<unsigned long long>(out) & 15) == 0;
  // the covariance of u on the fast hardware functions (v_sqrt_f32, v_exp_f32: ~1 ulp each; the fp32 Gram is held to 2e-5)
  auto cov = [&](float u) -> float {
    if (KID == HBO_KERNEL_SE) return sv * __builtin_amdgcn_exp2f(u * (-0.5f * 1.44269504088896341f));
    const float r = __builtin_amdgcn_sqrtf((KID == HBO_KERNEL_MATERN32 ? 3.f : 5.f) * u);
    const float e = sv * __builtin_amdgcn_exp2f(r * -1.44269504088896341f);
    return KID == HBO_KERNEL_MATERN32 ? e * (1.f + r) : e * fmaf(r, fmaf(r, 1.f / 3.f, 1.f), 1.f);
  };
  // accumulator layout of v_mfma_f32_32x32x16_bf16 with the operands swapped: Gram row = lane & 31 of block b, Gram column =
  // (q & 3) + 8 (q >> 2) + 4 (lane >> 5) of block a
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const int rl = wm * 64 + b * 32 + l32;
    const float na = sN[0][rl];
    const int64_t gr = r0 + rl;
    float* orow = out + gr * ldo + c0;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      __builtin_amdgcn_sched_barrier(0);   // (one block of 16 elements at a time)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int cl = wn * 64 + a * 32 + 8 * j + 4 * lh;
        const float4 nb4 = *reinterpret_cast<const float4*>(&sN[1][cl]);
        const float nbv[4] = {nb4.x, nb4.y, nb4.z, nb4.w};
        float val[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float u = fmaxf(na + nbv[e] - 2.f * acc[a][b][4 * j + e], 0.f);
          if (interior) { val[e] = cov(u); continue; }
          const int64_t gcol = c0 + cl + e;
          if (on_diag && gr == gcol) u = 0.f;   // a point and itself: exactly sv (+ noise + jitter)
          if (gr < n1 && gcol < n2) {
            val[e] = cov(u);
            if (on_diag && gr == gcol) val[e] += diag_add;
          } else {
            val[e] = (g.symmetric && gr == gcol) ? 1.f : 0.f;   // identity / zero padding
          }
        }
        if (interior) {
          { V16<float>::type vv; vv[0] = val[0]; vv[1] = val[1]; vv[2] = val[2]; vv[3] = val[3]; gst(reinterpret_cast<V16<float>::type*>(orow + cl), vv); }
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (gr < e1 && c0 + cl + e < e2) gst(orow + cl + e, val[e]);
        }
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
// hbo_tune gram_mfma (process-wide): fp32 Gram matrices of the stationary covariances with at least this many features take
// gram_mfma_kernel (0: never)
int g_gram_mfma_min_f = 32;
template <typename T>
void launch_gram_t(const GramArgs& a, const ModelDev* md, dim3 grid, hipStream_t st) {
  if constexpr (sizeof(T) == 4) {
    if (g_gram_mfma_min_f > 0 && a.fdim >= g_gram_mfma_min_f && a.kernel_id != HBO_KERNEL_DOT && !a.direct_form) {
      switch (a.kernel_id) {
        case HBO_KERNEL_SE: hipLaunchKernelGGL((gram_mfma_kernel<HBO_KERNEL_SE>), grid, dim3(256), 0, st, a, md); break;
        case HBO_KERNEL_MATERN32: hipLaunchKernelGGL((gram_mfma_kernel<HBO_KERNEL_MATERN32>), grid, dim3(256), 0, st, a, md); break;
        default: hipLaunchKernelGGL((gram_mfma_kernel<HBO_KERNEL_MATERN52>), grid, dim3(256), 0, st, a, md); break;
      }
      return;
    }
  }
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

void gram_set_mfma_min_features(int f) { g_gram_mfma_min_f = f; }
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