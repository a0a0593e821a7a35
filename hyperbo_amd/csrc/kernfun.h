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

// exp for the pair kernels (hyperbo/gp_utils/kernel.py:63-123: exp(-u/2), exp(-sqrt(3u)), exp(-sqrt(5u))): the Gram build and the
// gradient contraction are bound by their 32 / 64 fp64 exponentials per thread, and the library exp spends half of its ~40
// instructions on special cases these arguments never hit.  Cody-Waite reduction x = k ln2 + r (|r| <= 0.3466, ln2 split so that
// k * ln2_hi is exact), degree-13 Taylor polynomial in Horner form (truncation 4e-18), v_ldexp_f64 for 2^k: < 1.5 ulp on the
// arguments the kernels produce (tests/test_gpu_parity.py::test_device_exp_against_numpy), ~20 instructions.  Arguments below -746
// give 0 through the ldexp underflow; NaN stays NaN (the clamp is two compares, which a NaN fails; fmax / fmin would swallow it);
// large positive arguments are clamped (the kernels only pass x <= 0).
// Coefficients 1/13! ... 1/0! of the polynomial, held in VGPRs for the whole epilogue of a kernel (hbo_exp_coef once per thread):
// as literals they sit in scalar registers, and the compiler then evaluates Horner's p = p r + c with the two-address v_fmac_f64,
// whose accumulator must first be LOADED with c -- two v_mov_b32 per step, a fifth of the Gram kernel's instructions.  Against
// live vector registers it takes the three-address v_fma_f64.
struct ExpCoef { double c[14]; };
__device__ __forceinline__ ExpCoef hbo_exp_coef() {
  ExpCoef e = {{1.0 / 6227020800.0, 1.0 / 479001600.0, 1.0 / 39916800.0, 1.0 / 3628800.0, 1.0 / 362880.0, 1.0 / 40320.0, 1.0 / 5040.0,
                1.0 / 720.0, 1.0 / 120.0, 1.0 / 24.0, 1.0 / 6.0, 0.5, 1.0, 1.0}};
#pragma unroll
  for (int i = 0; i < 14; ++i) asm volatile("" : "+v"(e.c[i]));   // opaque: not folded back into literals
  return e;
}
// (the same polynomial with literal coefficients: kernels that have no 28 registers to spare for them, see grad_contract_kernel)
struct ExpLit {
  static constexpr double c[14] = {1.0 / 6227020800.0, 1.0 / 479001600.0, 1.0 / 39916800.0, 1.0 / 3628800.0, 1.0 / 362880.0, 1.0 / 40320.0, 1.0 / 5040.0,
                                   1.0 / 720.0, 1.0 / 120.0, 1.0 / 24.0, 1.0 / 6.0, 0.5, 1.0, 1.0};
};
template <typename C>
__device__ __forceinline__ double hbo_exp(double x, const C& ec) {
  x = x < -746.0 ? -746.0 : (x > 709.0 ? 709.0 : x);
  const double k = rint(x * 1.4426950408889634074);
  double r = fma(k, -6.93147180369123816490e-01, x);
  r = fma(k, -1.90821492927058770002e-10, r);
  double p = fma(ec.c[0], r, ec.c[1]);
#pragma unroll
  for (int i = 2; i < 14; ++i) p = fma(p, r, ec.c[i]);
  return ldexp(p, (int)k);
}
template <typename C>
__device__ __forceinline__ float hbo_exp(float x, const C&) { return expf(x); }

template <typename T, typename C>
__device__ __forceinline__ T kfun(int kid, T acc, T sv, T inv_sigma2, T bias2, const C& ec) {
  switch (kid) {
    case HBO_KERNEL_SE: return sv * hbo_exp((T)-0.5 * acc, ec);
    case HBO_KERNEL_MATERN32: { T r = sqrt((T)3 * acc); return sv * ((T)1 + r) * hbo_exp(-r, ec); }
    case HBO_KERNEL_MATERN52: { T r = sqrt((T)5 * acc); return sv * ((T)1 + r + r * r / (T)3) * hbo_exp(-r, ec); }
    default: return acc * inv_sigma2 + bias2;
  }
}
// d k / d u (u = scaled squared distance); 0 where u == 0 for Matern (linalg.py:183-188)
template <typename T, typename C>
__device__ __forceinline__ T dk_du(int kid, T u, T k, T sv, const C& ec) {
  switch (kid) {
    case HBO_KERNEL_SE: return (T)-0.5 * k;
    case HBO_KERNEL_MATERN32: { T r = sqrt((T)3 * u); return u == (T)0 ? (T)0 : -sv * (T)1.5 * hbo_exp(-r, ec); }
    case HBO_KERNEL_MATERN52: { T r = sqrt((T)5 * u); return u == (T)0 ? (T)0 : -sv * ((T)5 / (T)6) * hbo_exp(-r, ec) * ((T)1 + r); }
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
  const bool dok = d < fdim;
  const T sc = (scale && dok) ? (T)inv_ls[d] : (T)1;
  // every load is issued -- from a clamped, always valid address -- before the first one is waited for: with the bounds test around the
  // load the compiler put each in a branch of its own with a full wait behind it (8 L2 round trips in a row per operand and chunk)
  const int dc = dok ? d : 0;
  T v[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int64_t row = r0 + rr0 + 16 * q;
    const int64_t rc = row < n ? row : (n > 0 ? n - 1 : 0);   // (callers never come here with n = 0: empty sub-datasets are skipped)
    v[q] = gld(x + rc * fdim + dc);
  }
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int rr = rr0 + 16 * q;
    s[dd * SXS + rr] = (r0 + rr < n && dok) ? v[q] * sc : (T)0;
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
