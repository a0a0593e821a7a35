// fp32 posterior product V = W Kxq on the fp16 matrix cores from TWO-way splits ("f16x2"), gfx950.
//
// Same product as post3.hip (hyperbo/gp_utils/gp.py:295-305: the predictive variance needs V = L^-1 Kxq, N^2 M flops), half the
// matrix-core work.  post3's exact bf16 split needs three pieces per operand (8 significand bits each) and six MFMAs per product;
// fp16 carries 11 bits, so TWO pieces hold 22 of an fp32 number's 24 significand bits:
//     x s = h + l + r,   h = fp16(x s),  l = fp16(x s - h),  |r| <= 2^-22 |x s|   (s: a power of two that maps the operand's
//     largest magnitude to [2^13, 2^14) -- fp16's exponent range is narrow; entries below 2^-14 of that lose bits of l, i.e.
//     carry an ABSOLUTE error <= 2^-25 against a largest entry of 2^13: 4e-12 relative to the operand's scale)
// and the three products h h' + h l' + l h' (each exact in the fp32 accumulator: 11 x 11 bits) drop terms of relative weight
// 2^-22 per product.  That is a representation error of 2.4e-7 per term -- not exact like bf16x3, but below what the fp32
// accumulation of a K = 16 384 dot product commits anyway (~sqrt(K) 2^-24 = 7.6e-6), which is why the result is as close to
// the fp64 posterior as the fp32-MFMA product is (tests/test_gpu_parity.py::test_fp32_posterior_on_f16x2...).  Three
// v_mfma_f32_32x32x16_f16 per 16 values of k instead of six bf16 ones: the kernel is bound by the power the matrix cores draw
// (post3: 62 % MFMA-busy at 1.56 GHz), so half the MFMAs is most of the time.
// Used for the stationary covariances (|k(x, x')| <= signal variance gives the cross-Gram's scale without a pass over it);
// the dot-product kernel keeps bf16x3.
#include "hbo_internal.h"
#include <algorithm>

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned short u16;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct alignas(16) U16x8 { u16 v[8]; };
constexpr int P2_CHUNK = HBO_TILE * 16;   // elements of one plane of one block

// max |W| over the lower triangle (by 128-blocks) -> *out (as the bits of a non-negative float: integer max orders them)
__global__ __launch_bounds__(256) void absmax_lower_kernel(const float* __restrict__ in, int64_t ld, unsigned int* out) {
  const int R = blockIdx.y, C = blockIdx.x;
  if (C > R) return;
  float m = 0.f;
  for (int r = threadIdx.x >> 5; r < HBO_TILE; r += 8) {
    const float4 v = *reinterpret_cast<const float4*>(in + (int64_t)(R * HBO_TILE + r) * ld + C * HBO_TILE + (threadIdx.x & 31) * 4);
    m = fmaxf(m, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
    // (a NaN entry: fmaxf drops it here; the product then carries the NaN through its planes all the same)
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0) atomicMax(out, __float_as_uint(m));
}

// Layout of a split operand: as post3.hip's panel blocks with TWO planes:
//     element (row, k, plane p)  ->  ((R * nkb + KB) * 2 + p) * 2048 + (row % 128) * 16 + k % 16
// rows x ld fp32, k = column; only blocks up to the row tile's diagonal block (W is lower triangular); scale from *amax_bits
__global__ __launch_bounds__(256) void split2h_rows_kernel(const float* __restrict__ in, int64_t ld, u16* __restrict__ out, int nkb,
                                                           const unsigned int* __restrict__ amax_bits) {
  const int R = blockIdx.y, kb0 = blockIdx.x * 4;
  if (kb0 >= (R + 1) * (HBO_TILE / 16)) return;
  const float s = hbo_h2_scale_for(__uint_as_float(*amax_bits));
  const int row = threadIdx.x >> 1, half = threadIdx.x & 1;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int kb = kb0 + q;
    const float* src = in + (int64_t)(R * HBO_TILE + row) * ld + kb * 16 + half * 8;
    const float4 a = *reinterpret_cast<const float4*>(src);
    const float4 b = *reinterpret_cast<const float4*>(src + 4);
    const float x[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    U16x8 h, l;
#pragma unroll
    for (int e = 0; e < 8; ++e) hbo_split2h(x[e] * s, h.v[e], l.v[e]);
    u16* o = out + ((int64_t)R * nkb + kb) * 2 * P2_CHUNK + threadIdx.x * 8;
    *reinterpret_cast<U16x8*>(o) = h;
    *reinterpret_cast<U16x8*>(o + P2_CHUNK) = l;
  }
}
// krows x ld fp32 with k = ROW (the cross Gram: k = training point, column j = candidate) -> panel blocks of the transpose
__global__ __launch_bounds__(256) void split2h_transpose_kernel(const float* __restrict__ in, int64_t ld, u16* __restrict__ out, int nkb, float s) {
  __shared__ float tile[64][65];
  const int k0 = blockIdx.y * 64, j0 = blockIdx.x * 64;
  const int tid = threadIdx.x;
  {
    const int c = (tid & 15) * 4, r = tid >> 4;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 v = *reinterpret_cast<const float4*>(in + (int64_t)(k0 + r + 16 * q) * ld + j0 + c);
      tile[r + 16 * q][c] = v.x; tile[r + 16 * q][c + 1] = v.y; tile[r + 16 * q][c + 2] = v.z; tile[r + 16 * q][c + 3] = v.w;
    }
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int item = tid + 256 * q;
    const int kbl = item >> 7, j = (item & 127) >> 1, half = item & 1;
    const int ko = kbl * 16 + half * 8;
    U16x8 h, l;
#pragma unroll
    for (int e = 0; e < 8; ++e) hbo_split2h(tile[ko + e][j] * s, h.v[e], l.v[e]);
    const int jr = j0 + j;
    u16* o = out + ((int64_t)(jr / HBO_TILE) * nkb + (k0 / 16 + kbl)) * 2 * P2_CHUNK + (jr % HBO_TILE) * 16 + half * 8;
    *reinterpret_cast<U16x8*>(o) = h;
    *reinterpret_cast<U16x8*>(o + P2_CHUNK) = l;
  }
}

// ---- the product: post3_kernel's pipeline (128 x 128 tile, four waves 2 x 2 of 64 x 64, one stage = 16 values of k, swizzled
// 32-byte LDS rows, global loads four stages ahead in registers) on two planes per operand and three MFMAs per pair of fragments
constexpr int P2_ROW = 32;
constexpr int P2_ARR = 128 * P2_ROW;
constexpr int POST2H_LDS_BYTES = 2 * 2 * 2 * P2_ARR;   // stages x operands x planes: 32 KB

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void post2h_kernel(Post2hArgs g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float inv_scale = 1.f / (hbo_h2_scale_for(__uint_as_float(*g.wmax_bits)) * g.kscale);
  __shared__ int s_tile;
  for (int tile = (int)blockIdx.y * (int)gridDim.x + (int)blockIdx.x;;) {
  if (g.work_counter) {
    if (tid == 0) s_tile = atomicAdd(g.work_counter, 1);
    __syncthreads();
    tile = s_tile;
    __syncthreads();
    if (tile >= g.col_tiles * g.nblk) break;
  }
  const int i = g.nblk - 1 - tile / g.col_tiles;   // long rows first
  const int jq = tile % g.col_tiles;
  const int wm = wave >> 1, wn = wave & 1;
  const int l32 = lane & 31, lh = lane >> 5;
  auto arr = [&](int st, int op, int p) { return smem + (size_t)((st * 2 + op) * 2 + p) * P2_ARR; };

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  const int srow = tid >> 1, shalf = tid & 1;
  const u16* ga = g.Wp + (int64_t)i * g.nkb * 2 * P2_CHUNK + tid * 8;
  const u16* gb = g.Kp + (int64_t)jq * g.nkb * 2 * P2_CHUNK + tid * 8;
  const int soff = srow * P2_ROW + ((shalf ^ ((srow >> 3) & 1)) * 16);
  struct Slot { u32x4 a0, a1, b0, b1; };
  Slot s0, s1, s2, s3;
#define P2_GLOAD(KT, S)                                                              \
  {                                                                                  \
    const u16* pa_ = ga + (int64_t)(KT) * 2 * P2_CHUNK;                              \
    const u16* pb_ = gb + (int64_t)(KT) * 2 * P2_CHUNK;                              \
    S.a0 = *reinterpret_cast<const u32x4*>(pa_);                                     \
    S.b0 = *reinterpret_cast<const u32x4*>(pb_);                                     \
    S.a1 = *reinterpret_cast<const u32x4*>(pa_ + P2_CHUNK);                          \
    S.b1 = *reinterpret_cast<const u32x4*>(pb_ + P2_CHUNK);                          \
  }
#define P2_SSTORE(ST, S)                                                             \
  {                                                                                  \
    *reinterpret_cast<u32x4*>(arr(ST, 0, 0) + soff) = S.a0;                          \
    *reinterpret_cast<u32x4*>(arr(ST, 1, 0) + soff) = S.b0;                          \
    *reinterpret_cast<u32x4*>(arr(ST, 0, 1) + soff) = S.a1;                          \
    *reinterpret_cast<u32x4*>(arr(ST, 1, 1) + soff) = S.b1;                          \
  }
#define P2_STAGE(KT, CUR, S_FILL, S_NEXT)                                                                         \
  {                                                                                                               \
    const int kt_ = (KT);                                                                                         \
    if (kt_ + 4 < nk) P2_GLOAD(kt_ + 4, S_FILL)                                                                   \
    f16x8 fa[2][2], fb[2][2];                                                                                     \
    _Pragma("unroll") for (int p = 0; p < 2; ++p)                                                                 \
    _Pragma("unroll") for (int t = 0; t < 2; ++t) {                                                               \
      fa[p][t] = *reinterpret_cast<const f16x8*>(arr(CUR, 0, p) + foff_a + t * 32 * P2_ROW);                      \
      fb[p][t] = *reinterpret_cast<const f16x8*>(arr(CUR, 1, p) + foff_b + t * 32 * P2_ROW);                      \
    }                                                                                                             \
    constexpr int PA[3] = {1, 0, 0};   /* smallest products first: l h', h l', h h' */                            \
    constexpr int PB[3] = {0, 1, 0};                                                                              \
    _Pragma("unroll") for (int q = 0; q < 3; ++q)                                                                 \
    _Pragma("unroll") for (int a = 0; a < 2; ++a)                                                                 \
    _Pragma("unroll") for (int b = 0; b < 2; ++b)                                                                 \
      acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[PA[q]][a], fb[PB[q]][b], acc[a][b], 0, 0, 0);         \
    if (kt_ + 1 < nk) P2_SSTORE((CUR) ^ 1, S_NEXT)                                                                \
    __syncthreads();                                                                                              \
  }
  const int nk = (i + 1) * HBO_TILE / 16;   // a multiple of 8
  const int fsw = (lh ^ ((l32 >> 3) & 1)) * 16;
  const int foff_a = (wm * 64 + l32) * P2_ROW + fsw;
  const int foff_b = (wn * 64 + l32) * P2_ROW + fsw;
  P2_GLOAD(0, s0) P2_GLOAD(1, s1) P2_GLOAD(2, s2) P2_GLOAD(3, s3)
  P2_SSTORE(0, s0)
  __syncthreads();
  for (int kt0 = 0; kt0 < nk; kt0 += 4) {
    P2_STAGE(kt0, 0, s0, s1)
    P2_STAGE(kt0 + 1, 1, s1, s2)
    P2_STAGE(kt0 + 2, 0, s2, s3)
    P2_STAGE(kt0 + 3, 1, s3, s0)
  }
#undef P2_STAGE
#undef P2_GLOAD
#undef P2_SSTORE
  if (g.colsq) {
    float* red = reinterpret_cast<float*>(smem);   // [4 waves][64]  (the k loop ended with a barrier)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      float s = 0.f;
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) { const float v = acc[a][b][r] * inv_scale; s += v * v; }
      s += __shfl_xor(s, 32);
      if (lh == 0) red[wave * 64 + b * 32 + l32] = s;
    }
    __syncthreads();
    if (tid < 128) {
      const int wn2 = tid >> 6, c = tid & 63;
      g.colsq[(int64_t)i * g.ldc + (int64_t)jq * HBO_TILE + tid] = red[(0 * 2 + wn2) * 64 + c] + red[(1 * 2 + wn2) * 64 + c];
    }
  }
  if (!g.work_counter) break;
  __syncthreads();
  }
}
}  // namespace

void launch_absmax_lower(const float* in, int64_t ld, int nblk, unsigned int* out, hipStream_t st) {
  (void)hipMemsetAsync(out, 0, sizeof(unsigned int), st);
  hipLaunchKernelGGL(absmax_lower_kernel, dim3(nblk, nblk), dim3(256), 0, st, in, ld, out);
}
void launch_split2h_rows(const float* in, int64_t ld, int row_tiles, unsigned short* out, int nkb, const unsigned int* amax_bits, hipStream_t st) {
  hipLaunchKernelGGL(split2h_rows_kernel, dim3((nkb + 3) / 4, row_tiles), dim3(256), 0, st, in, ld, out, nkb, amax_bits);
}
void launch_split2h_transpose(const float* in, int64_t ld, int krows, int jcols, unsigned short* out, int nkb, float scale, hipStream_t st) {
  if (krows <= 0 || jcols <= 0) return;
  hipLaunchKernelGGL(split2h_transpose_kernel, dim3(jcols / 64, krows / 64), dim3(256), 0, st, in, ld, out, nkb, scale);
}
// power of two that maps a bound on |Kxq| into [2^13, 2^14): the host-side twin of hbo_h2_scale_for
float post2h_scale_for(double bound) {
  if (!(bound > 0) || !(bound < 1e30)) return 1.f;
  int e;
  (void)frexp(bound, &e);
  return ldexpf(1.f, 14 - e);
}
void launch_post2h(const Post2hArgs& a_in, int col_tiles, hipStream_t st) {
  static unsigned long long seen = 0;
  if (hbo_first_use_on_device(seen))
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&post2h_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, POST2H_LDS_BYTES);
  Post2hArgs a = a_in; a.col_tiles = col_tiles;
  if (a.work_counter) {
    int dev = 0, cus = 256;
    hipGetDevice(&dev); hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const int resident = 2 * cus;
    if (col_tiles * a.nblk > 2 * resident) { hipLaunchKernelGGL(post2h_kernel, dim3(resident, 1), dim3(256), POST2H_LDS_BYTES, st, a); return; }
    a.work_counter = nullptr;
  }
  hipLaunchKernelGGL(post2h_kernel, dim3(col_tiles, a.nblk), dim3(256), POST2H_LDS_BYTES, st, a);
}
