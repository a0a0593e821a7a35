// Device helpers shared by the elementwise / reduction kernels of the GP hot path (gram.hip, grad.hip, post.hip): the pair
// kernels of hyperbo/gp_utils/kernel.py:63-145 and their derivatives, LDS staging of feature blocks, the means of
// hyperbo/gp_utils/mean.py:30-79, block reductions, the normal pdf / cdf of the acquisition functions (acfun.py:96-142).
#pragma once
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

// Gram / contraction tile = 64 rows x 128 columns per 256-thread workgroup, 4x8 register micro-tile per thread
// (an 8x8 micro-tile needs 256 VGPRs in fp64 -> 1 wave/SIMD and exposed exp/store latency).
constexpr int GRA = 4;            // rows per thread
constexpr int GTR = 16 * GRA;     // tile rows
__device__ __forceinline__ double norm_pdf(double x) { return exp(-0.5 * x * x) * 0.3989422804014327; }
__device__ __forceinline__ double norm_cdf(double x) { return 0.5 * erfc(-x * 0.7071067811865476); }
}  // namespace
