// fp32 posterior product V = W Kxq on the bf16 matrix cores ("bf16x3"), gfx950.
//
// hyperbo/gp_utils/gp.py:295-305 solves L V = Kxq (solve_triangular) for the predictive variance; here W = L^-1 is cached and
// V = W Kxq is a triangular GEMM of N^2 M flops -- at the reference's default dtype (fp32) 1.76e13 flops at cfg 3, bound by the
// fp32 MFMA rate (157 TFLOP/s dense).  The bf16 MFMA runs at 16x that rate, and an fp32 number is EXACTLY the sum of three
// bf16 numbers (24 significand bits = 8 + 8 + 8, same exponent range):  x = x0 + x1 + x2,  x0 = bf16(x), x1 = bf16(x - x0),
// x2 = bf16(x - x0 - x1).  With both operands split, the six products of weight >= 2^-16 relative to the leading one,
//     a0 b0 + (a0 b1 + a1 b0) + (a0 b2 + a1 b1 + a2 b0),
// are formed exactly by the bf16 MFMA (8 x 8 bit products) and accumulated in its fp32 accumulator; the three dropped ones
// weigh <= 2^-24 -- the rounding an fp32 FMA commits on every product anyway.  Six bf16 MFMAs (32 cycles each for 32x32x16)
// replace eight fp32 MFMAs (64 cycles each for 32x32x2): 2.7x the rate at fp32 accuracy (measured against the fp64 path in
// tests/test_gpu_parity.py).  Operands are split once: W per factorisation (split3_rows), the cross-Gram per candidate chunk,
// transposed on the way so that both operands are k-contiguous (split3_transpose), into a blocked layout that makes every
// pipeline stage of the product a contiguous read.
#include "hbo_internal.h"
#include <algorithm>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned short u16;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));   // (native vector: HIP's uint4 is a union-backed struct that kept the staging registers in scratch memory)

__device__ __forceinline__ u16 bf_bits(__bf16 v) { return __builtin_bit_cast(u16, v); }

struct alignas(16) U16x8 { u16 v[8]; };

// What a split kernel does with eight consecutive k values of one operand row: MODE 0 three bf16 planes (exact), MODE 1 two fp16
// planes of the values scaled by the power of two `sc` (hbo_split2h), MODE 2 nothing but their largest magnitude (the pass that
// finds a MODE 1 split's scale walks exactly the same elements).  `o`: element (row, 8 half) of plane 0 of the block
template <int MODE> constexpr int planes_of() { return MODE == 1 ? 2 : 3; }
template <int MODE>
__device__ __forceinline__ void emit8(const float (&x)[8], float sc, u16* o, float& mx) {
  if constexpr (MODE == 2) {
#pragma unroll
    for (int e = 0; e < 8; ++e) mx = fmaxf(mx, fabsf(x[e]));
  } else if constexpr (MODE == 1) {
    U16x8 h, l;
#pragma unroll
    for (int e = 0; e < 8; ++e) hbo_split2h(x[e] * sc, h.v[e], l.v[e]);
    *reinterpret_cast<U16x8*>(o) = h;
    *reinterpret_cast<U16x8*>(o + HBO_TILE * 16) = l;
  } else {
    U16x8 h, m, l;
#pragma unroll
    for (int e = 0; e < 8; ++e) hbo_split3(x[e], h.v[e], m.v[e], l.v[e]);
    *reinterpret_cast<U16x8*>(o) = h;
    *reinterpret_cast<U16x8*>(o + HBO_TILE * 16) = m;
    *reinterpret_cast<U16x8*>(o + 2 * HBO_TILE * 16) = l;
  }
}
// MODE 2: the workgroup's maximum into *out (bits of a non-negative float: integer max orders them)
__device__ __forceinline__ void publish_max(float mx, unsigned int* out) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
  if ((threadIdx.x & 63) == 0 && mx > 0.f) atomicMax(out, __float_as_uint(mx));
}

// Layout of a split operand ("panel blocks"): for every 128-row tile R and every block KB of 16 values of k, the three planes
// of the 128 x 16 block are stored back to back, each as [row][16 k] (4 KB):
//     element (row, k, plane p)  ->  ((R * nkb + KB) * 3 + p) * 2048 + (row % 128) * 16 + k % 16,   nkb = Kpad / 16.
// One pipeline stage of the product kernel is then 3 x 4 KB of contiguous memory per operand (256 threads x 16 bytes per
// plane): the first version kept row-major planes and fetched 32 bytes out of every 128-byte line per stage (40 TFLOP/s).
constexpr int P3_CHUNK = HBO_TILE * 16;   // elements of one plane of one block

// in: rows x ld fp32 (row-major, k = column).  One workgroup = one 128-row tile x four k blocks; only blocks up to the row
// tile's own diagonal block are written (W is lower triangular, zeros above the diagonal inside the diagonal blocks).
__global__ __launch_bounds__(256) void split3_rows_kernel(const float* __restrict__ in, int64_t ld, u16* __restrict__ out, int nkb) {
  const int R = blockIdx.y, kb0 = blockIdx.x * 4;
  if (kb0 >= (R + 1) * (HBO_TILE / 16)) return;
  const int row = threadIdx.x >> 1, half = threadIdx.x & 1;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int kb = kb0 + q;
    const float* src = in + (int64_t)(R * HBO_TILE + row) * ld + kb * 16 + half * 8;
    const float4 a = *reinterpret_cast<const float4*>(src);
    const float4 b = *reinterpret_cast<const float4*>(src + 4);
    const float x[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    U16x8 h, m, l;
#pragma unroll
    for (int e = 0; e < 8; ++e) hbo_split3(x[e], h.v[e], m.v[e], l.v[e]);
    u16* o = out + ((int64_t)R * nkb + kb) * 3 * P3_CHUNK + threadIdx.x * 8;
    *reinterpret_cast<U16x8*>(o) = h;
    *reinterpret_cast<U16x8*>(o + P3_CHUNK) = m;
    *reinterpret_cast<U16x8*>(o + 2 * P3_CHUNK) = l;
  }
}

// in: krows x ld fp32 with k = ROW (the cross-Gram Kxq: k = training point, column j = candidate); out: panel blocks of the
// transpose (row = j).  64 (k) x 64 (j) tiles through LDS.
template <int MODE>
__global__ __launch_bounds__(256) void split3_transpose_kernel(const float* __restrict__ in, int64_t ld, u16* __restrict__ out, int nkb, int lower_only,
                                                               const unsigned int* scale_bits, unsigned int* max_out) {
  constexpr int NP = planes_of<MODE>();
  __shared__ float tile[64][65];
  const int k0 = blockIdx.y * 64, j0 = blockIdx.x * 64;
  if (lower_only && k0 < j0 / HBO_TILE * HBO_TILE) return;   // `in` lower triangular by 128-blocks: nobody reads the blocks above
  const int tid = threadIdx.x;
  const float sc = MODE == 1 ? hbo_h2_scale_for(__uint_as_float(*scale_bits)) : 1.f;
  float mx = 0.f;
  {
    const int c = (tid & 15) * 4, r = tid >> 4;   // 16 threads x float4 per row of 64, 16 rows per pass
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 v = *reinterpret_cast<const float4*>(in + (int64_t)(k0 + r + 16 * q) * ld + j0 + c);
      if constexpr (MODE == 2) mx = fmaxf(mx, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
      else { tile[r + 16 * q][c] = v.x; tile[r + 16 * q][c + 1] = v.y; tile[r + 16 * q][c + 2] = v.z; tile[r + 16 * q][c + 3] = v.w; }
    }
  }
  if constexpr (MODE == 2) { publish_max(mx, max_out); return; }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int item = tid + 256 * q;          // 4 k blocks x 64 rows j x 2 halves: consecutive items -> consecutive 16 bytes
    const int kbl = item >> 7, j = (item & 127) >> 1, half = item & 1;
    const int ko = kbl * 16 + half * 8;
    float x[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = tile[ko + e][j];
    const int jr = j0 + j;
    u16* o = out + ((int64_t)(jr / HBO_TILE) * nkb + (k0 / 16 + kbl)) * NP * P3_CHUNK + (jr % HBO_TILE) * 16 + half * 8;
    emit8<MODE>(x, sc, o, mx);
  }
}

// ---- the product ---------------------------------------------------------------------------------------------------------
// One workgroup = one 128 x 128 tile of V (rows i of W, columns j of the chunk), four waves in 2 x 2, each 64 x 64 = 2 x 2 MFMA
// tiles of 32 x 32 (64 accumulator registers).  One pipeline stage = 16 values of k = one MFMA depth: per operand and plane
// 128 rows x 32 bytes in LDS, unpadded, the two 16-byte halves of a row swapped in every other group of eight rows: the
// 16-lane groups of ds_read_b128 ({0-3,12-15,20-27}, ... -- MI355X_MICROARCH.md, LDS) and the 8-lane groups of
// ds_write_b128 then cover all banks exactly once (a 48-byte row stride was conflict-free for the reads only: a third of the
// LDS cycles were write conflicts, SQ_LDS_BANK_CONFLICT).  Two stages of 2 operands x 3 planes: 48 KB.
constexpr int P3_ROW = 32;                     // bytes per LDS row
constexpr int P3_ARR = 128 * P3_ROW;           // one operand plane of one stage
constexpr int POST3_LDS_BYTES = 2 * 2 * 3 * P3_ARR;

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void post3_kernel(Post3Args g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // one tile per workgroup, or (work_counter) a resident grid drawing the tiles in the same order -- long rows first -- from a
  // counter: the hardware deals a plain grid's workgroups to the 8 XCDs in turn and waits when the next one's XCD is full
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
  auto arr = [&](int st, int op, int p) { return smem + (size_t)((st * 2 + op) * 3 + p) * P3_ARR; };

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  // staging: thread -> (row, 16-byte half of the stage's 32 bytes)
  const int srow = tid >> 1, shalf = tid & 1;
  const u16* ga = g.Wp + (int64_t)i * g.nkb * 3 * P3_CHUNK + tid * 8;
  const u16* gb = g.Kp + (int64_t)jq * g.nkb * 3 * P3_CHUNK + tid * 8;
  const int soff = srow * P3_ROW + ((shalf ^ ((srow >> 3) & 1)) * 16);
  // Global loads run four stages ahead of their use, in registers: one stage is only 24 MFMAs per wave (768 cycles, 0.3 us),
  // far less than a memory round trip -- with a single stage in flight the kernel ran at the latency of its loads (72 TFLOP/s).
  struct Slot { u32x4 a0, a1, a2, b0, b1, b2; };
  Slot s0, s1, s2, s3;   // (named, not an array: an array indexed through the unrolled loop ended up in scratch memory)
#define P3_GLOAD(KT, S)                                                              \
  {                                                                                  \
    const u16* pa_ = ga + (int64_t)(KT) * 3 * P3_CHUNK;                              \
    const u16* pb_ = gb + (int64_t)(KT) * 3 * P3_CHUNK;                              \
    S.a0 = *reinterpret_cast<const u32x4*>(pa_);                                     \
    S.b0 = *reinterpret_cast<const u32x4*>(pb_);                                     \
    S.a1 = *reinterpret_cast<const u32x4*>(pa_ + P3_CHUNK);                          \
    S.b1 = *reinterpret_cast<const u32x4*>(pb_ + P3_CHUNK);                          \
    S.a2 = *reinterpret_cast<const u32x4*>(pa_ + 2 * P3_CHUNK);                      \
    S.b2 = *reinterpret_cast<const u32x4*>(pb_ + 2 * P3_CHUNK);                      \
  }
#define P3_SSTORE(ST, S)                                                             \
  {                                                                                  \
    *reinterpret_cast<u32x4*>(arr(ST, 0, 0) + soff) = S.a0;                          \
    *reinterpret_cast<u32x4*>(arr(ST, 1, 0) + soff) = S.b0;                          \
    *reinterpret_cast<u32x4*>(arr(ST, 0, 1) + soff) = S.a1;                          \
    *reinterpret_cast<u32x4*>(arr(ST, 1, 1) + soff) = S.b1;                          \
    *reinterpret_cast<u32x4*>(arr(ST, 0, 2) + soff) = S.a2;                          \
    *reinterpret_cast<u32x4*>(arr(ST, 1, 2) + soff) = S.b2;                          \
  }
  // one pipeline stage: refill slot S_FILL (its data went to LDS one stage ago) with stage kt + 4, run the 24 MFMAs of the
  // stage in LDS buffer CUR, move slot S_NEXT (stage kt + 1, loaded three stages ago) into the other LDS buffer
#define P3_STAGE(KT, CUR, S_FILL, S_NEXT)                                                                         \
  {                                                                                                               \
    const int kt_ = (KT);                                                                                         \
    if (kt_ + 4 < nk) P3_GLOAD(kt_ + 4, S_FILL)                                                                   \
    bf16x8 fa[3][2], fb[3][2];                                                                                    \
    _Pragma("unroll") for (int p = 0; p < 3; ++p)                                                                 \
    _Pragma("unroll") for (int t = 0; t < 2; ++t) {                                                               \
      fa[p][t] = *reinterpret_cast<const bf16x8*>(arr(CUR, 0, p) + foff_a + t * 32 * P3_ROW);                     \
      fb[p][t] = *reinterpret_cast<const bf16x8*>(arr(CUR, 1, p) + foff_b + t * 32 * P3_ROW);                     \
    }                                                                                                             \
    constexpr int PA[6] = {2, 0, 1, 1, 0, 0};   /* smallest products first */                                     \
    constexpr int PB[6] = {0, 2, 1, 0, 1, 0};                                                                     \
    _Pragma("unroll") for (int q = 0; q < 6; ++q)                                                                 \
    _Pragma("unroll") for (int a = 0; a < 2; ++a)                                                                 \
    _Pragma("unroll") for (int b = 0; b < 2; ++b)                                                                 \
      acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[PA[q]][a], fb[PB[q]][b], acc[a][b], 0, 0, 0);        \
    if (kt_ + 1 < nk) P3_SSTORE((CUR) ^ 1, S_NEXT)                                                                \
    __syncthreads();                                                                                              \
  }
  const int nk = (i + 1) * HBO_TILE / 16;   // a multiple of 8
  const int fsw = (lh ^ ((l32 >> 3) & 1)) * 16;   // (row = 64 w + 32 t + l32: bit 3 of the row is bit 3 of l32)
  const int foff_a = (wm * 64 + l32) * P3_ROW + fsw;
  const int foff_b = (wn * 64 + l32) * P3_ROW + fsw;
  P3_GLOAD(0, s0) P3_GLOAD(1, s1) P3_GLOAD(2, s2) P3_GLOAD(3, s3)
  P3_SSTORE(0, s0)
  __syncthreads();
  for (int kt0 = 0; kt0 < nk; kt0 += 4) {
    P3_STAGE(kt0, 0, s0, s1)
    P3_STAGE(kt0 + 1, 1, s1, s2)
    P3_STAGE(kt0 + 2, 0, s2, s3)
    P3_STAGE(kt0 + 3, 1, s3, s0)
  }
#undef P3_STAGE
#undef P3_GLOAD
#undef P3_SSTORE
  if (g.colsq) {
    float* red = reinterpret_cast<float*>(smem);   // [4 waves][64]  (the k loop ended with a barrier)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      float s = 0.f;
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[a][b][r] * acc[a][b][r];
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
  __syncthreads();   // (the staging buffers and `red` are reused by the next tile)
  }
}


// ---- the same product for the fp32 trailing updates of the blocked Cholesky ---------------------------------------------
// C[r, c] -= P[r, :] P[c, :]^T over the K = 16 * nk panel columns of a group: both operands are row tiles of ONE split copy of
// the group's panels (split3_panel_kernel: rows = matrix rows below the group incl. the augmented tile-row, k = panel column),
// the tile is read-modified-written once at the end.  hyperbo/basics/linalg.py:29-33 (cholesky) at the reference's default
// dtype; the panel kernels (potf2, panel solve) stay in fp32 arithmetic.
template <bool H2>
__global__ __launch_bounds__(256) void split3_panel_kernel(Syrk3Args g) {
  // one workgroup = one 128-row tile x four k blocks of the panel columns [kcol0, kcol0 + 16 * nk_split)
  constexpr int NP = H2 ? 2 : 3;
  const TaskDesc& t = g.tasks[blockIdx.z];
  const int R = g.r_lo + (int)blockIdx.y;
  if (R > t.nblk) return;                       // (row tile nblk = the augmented tile-row)
  const float* in = static_cast<const float*>(t.A) + (int64_t)g.kcol0;
  u16* out = g.Xp + (int64_t)blockIdx.z * g.task_stride;
  const int row = threadIdx.x >> 1, half = threadIdx.x & 1;
  if ((int)blockIdx.x * 4 >= g.nk_split) return;   // (nk_split is a multiple of 8: a workgroup's four blocks exist or none does)
  float x[4][8];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int kb = (int)blockIdx.x * 4 + q;     // block inside the split range
    const float* src = in + (int64_t)(R * HBO_TILE + row) * t.ld + kb * 16 + half * 8;
    const float4 a = *reinterpret_cast<const float4*>(src);
    const float4 b = *reinterpret_cast<const float4*>(src + 4);
    x[q][0] = a.x; x[q][1] = a.y; x[q][2] = a.z; x[q][3] = a.w; x[q][4] = b.x; x[q][5] = b.y; x[q][6] = b.z; x[q][7] = b.w;
  }
  float sc = 1.f;
  if constexpr (H2) {
    sc = g.sx;
    if (R == t.nblk) {
      // the augmented tile-row: scale of these 16 rows x 64 panel columns from their own maximum (32 lanes = 16 rows x 2 halves)
      float m = 0.f;
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int e = 0; e < 8; ++e) m = fmaxf(m, fabsf(x[q][e]));
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
      sc = hbo_h2_scale_for(m);
      if ((threadIdx.x & 31) == 0)
        g.aug_scale[(int64_t)blockIdx.z * g.aug_stride + (int64_t)((g.kb_off >> 2) + (int)blockIdx.x) * 8 + (threadIdx.x >> 5)] = sc;
    }
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int kb = (int)blockIdx.x * 4 + q;
    u16* o = out + ((int64_t)R * g.nkb + g.kb_off + kb) * NP * P3_CHUNK + threadIdx.x * 8;
    U16x8 h, m, l;
    if constexpr (H2) {
#pragma unroll
      for (int e = 0; e < 8; ++e) hbo_split2h(x[q][e] * sc, h.v[e], m.v[e]);
      *reinterpret_cast<U16x8*>(o) = h;
      *reinterpret_cast<U16x8*>(o + P3_CHUNK) = m;
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) hbo_split3(x[q][e], h.v[e], m.v[e], l.v[e]);
      *reinterpret_cast<U16x8*>(o) = h;
      *reinterpret_cast<U16x8*>(o + P3_CHUNK) = m;
      *reinterpret_cast<U16x8*>(o + 2 * P3_CHUNK) = l;
    }
  }
}

// rows of `in` = operand rows.  tri: only the blocks up to the row tile's own diagonal block hold data (a lower-triangular operand);
// the others are written as zeros (the product kernels bound K by the structure and never read them, but the buffer is reused)
template <int MODE>
__global__ __launch_bounds__(256) void split3_block_kernel(Split3Block g) {
  constexpr int NP = planes_of<MODE>();
  const int R = blockIdx.y;
  const bool last = blockIdx.z == gridDim.z - 1;
  if (last && R >= g.last_rows) return;
  const float* in = g.in + (int64_t)blockIdx.z * g.gstep;
  u16* out = g.out + (int64_t)blockIdx.z * g.gstride;
  const int row = threadIdx.x >> 1, half = threadIdx.x & 1;
  const float sc = MODE == 1 ? (g.scale_bits ? hbo_h2_scale_for(__uint_as_float(*g.scale_bits)) : g.scale) : 1.f;
  float mx = 0.f;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int kb = (int)blockIdx.x * 4 + q;
    if (kb >= g.nkb || (last && kb * 16 >= g.last_krows)) break;
    if (g.tri && kb >= (R + 1) * (HBO_TILE / 16)) break;
    const float* src = in + (int64_t)(R * HBO_TILE + row) * g.ld + kb * 16 + half * 8;
    const float4 a = *reinterpret_cast<const float4*>(src);
    const float4 b = *reinterpret_cast<const float4*>(src + 4);
    const float x[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    emit8<MODE>(x, sc, out + ((int64_t)R * g.nkb + kb) * NP * P3_CHUNK + threadIdx.x * 8, mx);
  }
  if constexpr (MODE == 2) publish_max(mx, g.max_out);
}
// columns of `in` = operand rows (k = row of `in`): 64 (k) x 64 (j) tiles through LDS, as split3_transpose_kernel
template <int MODE>
__global__ __launch_bounds__(256) void split3_block_t_kernel(Split3Block g) {
  constexpr int NP = planes_of<MODE>();
  __shared__ float tile[64][65];
  const float* in = g.in + (int64_t)blockIdx.z * g.gstep;
  u16* out = g.out + (int64_t)blockIdx.z * g.gstride;
  const int k0 = blockIdx.y * 64, j0 = blockIdx.x * 64;
  const int tid = threadIdx.x;
  if (blockIdx.z == gridDim.z - 1 && (k0 >= g.last_krows || j0 >= g.last_rows * HBO_TILE)) return;
  const float sc = MODE == 1 ? (g.scale_bits ? hbo_h2_scale_for(__uint_as_float(*g.scale_bits)) : g.scale) : 1.f;
  float mx = 0.f;
  {
    const int c = (tid & 15) * 4, r = tid >> 4;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 v = *reinterpret_cast<const float4*>(in + (int64_t)(k0 + r + 16 * q) * g.ld + j0 + c);
      if constexpr (MODE == 2) mx = fmaxf(mx, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
      else { tile[r + 16 * q][c] = v.x; tile[r + 16 * q][c + 1] = v.y; tile[r + 16 * q][c + 2] = v.z; tile[r + 16 * q][c + 3] = v.w; }
    }
  }
  if constexpr (MODE == 2) { publish_max(mx, g.max_out); return; }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int item = tid + 256 * q;
    const int kbl = item >> 7, j = (item & 127) >> 1, half = item & 1;
    const int ko = kbl * 16 + half * 8;
    float x[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = tile[ko + e][j];
    const int jr = j0 + j;
    u16* o = out + ((int64_t)(jr / HBO_TILE) * g.nkb + (k0 / 16 + kbl)) * NP * P3_CHUNK + (jr % HBO_TILE) * 16 + half * 8;
    emit8<MODE>(x, sc, o, mx);
  }
}

// H2: the f16x2 form (Syrk3Args::h2) -- two fp16 planes per operand, three MFMAs per pair of fragments, result scaled back
template <bool H2>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void syrk3_kernel(Syrk3Args g) {
  constexpr int NP = H2 ? 2 : 3;
  constexpr int LDS_BYTES = 2 * 2 * NP * P3_ARR;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const TaskDesc& t = g.tasks[blockIdx.z];
  // (the leading dimension in a register: read through the descriptor reference inside the epilogue it was re-fetched -- a vector load
  //  with a full wait behind it -- before every one of a lane's 64 C elements, each then a serial pair of L2 round trips)
  const int64_t ldc = __builtin_amdgcn_readfirstlane((int)t.ld);
  // tile (r, c) of the trapezoid c in [c_lo, c_hi), r in [c, nblk]: linear index, column-major (consecutive workgroups share B)
  const int nrt = t.nblk + 1;
  const int chi = g.c_hi < t.nblk ? g.c_hi : t.nblk;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int* const mailbox = reinterpret_cast<int*>(smem + LDS_BYTES);
  const int* const yslot = g.yield_flag ? g.yield_flag + cu_token() : nullptr;
  int ytok = 0;
  if (g.yield_mark && tid == 0) ytok = yield_enter(g.yield_mark);
  for (int tix0 = blockIdx.x;; tix0 += gridDim.x) {
  int tix = tix0;
  if (g.persistent) {
    if (tid == 0) mailbox[0] = atomicAdd(g.work_counter, 1);
    __syncthreads();
    tix = mailbox[0];
    __syncthreads();
  }
  int c = g.c_lo, r = 0, nk = g.nk;
  const u16 *ga, *gb;
  float* C;
  float csign = -1.f;      // C = cold * cbeta + csign * acc
  bool cbeta = true;
  const float* augs = nullptr;   // H2, the augmented tile-row: its scales per (chunk of four k blocks, 16 rows)
  if (g.mode == 0) {
    while (c < chi && tix >= nrt - c) { tix -= nrt - c; ++c; }
    if (c >= chi) break;
    r = c + tix;
    const u16* xp = g.Xp + (int64_t)blockIdx.z * g.task_stride;
    ga = xp + ((int64_t)r * g.nkb + g.kb_off) * NP * P3_CHUNK + tid * 8;
    gb = xp + ((int64_t)c * g.nkb + g.kb_off) * NP * P3_CHUNK + tid * 8;
    C = static_cast<float*>(t.A) + (int64_t)r * HBO_TILE * t.ld + (int64_t)c * HBO_TILE;
    if (H2 && r == t.nblk) augs = g.aug_scale + (int64_t)blockIdx.z * g.aug_stride + (int64_t)(g.kb_off >> 2) * 8;
  } else if (g.mode == 3) {
    // K^-1 = W^T W, lower tiles (i, jt <= i): S[i, jt] = sum_{k >= i} W[k, i]^T W[k, jt]; both operands are row tiles of the
    // split TRANSPOSE of W (Xp), K blocks [8 i, nkb).  Row tile i slow = K descending (longest first), jt fast: consecutive
    // workgroups share the A operand
    int lin = tix;
    const int nb = t.nblk;
    if (lin >= nb * (nb + 1) / 2) break;
    int i = 0;
    while (lin > i) { lin -= i + 1; ++i; }
    const int kb0 = 8 * i;
    nk = 8 * nb - kb0;
    ga = g.Xp + ((int64_t)i * g.nkb + kb0) * NP * P3_CHUNK + tid * 8;
    gb = g.Xp + ((int64_t)lin * g.nkb + kb0) * NP * P3_CHUNK + tid * 8;
    C = static_cast<float*>(t.S) + (int64_t)i * HBO_TILE * t.ld + (int64_t)lin * HBO_TILE;
    csign = 1.f; cbeta = false;
  } else {
    // tiles in launch order: the index that fixes K slowest (longest first), the rows / columns of all groups fast
    const int s = g.s, ng = g.ngrp;
    const int rows_total = (ng - 1) * s + g.vlast;   // tile rows over all groups (the last group's lower half may be cut)
    if (tix >= rows_total * s) break;
    int grp, it, jt;
    if (g.mode == 1) {           // K = 8 (s - jt) blocks: jt slow, ascending
      jt = tix / rows_total;
      const int rr = tix % rows_total;
      grp = rr / s < ng - 1 ? rr / s : ng - 1;
      it = rr - grp * s;
    } else {                     // K = 8 (it + 1) blocks: it slow, descending; every group but a cut last one has the row
      jt = tix % s;
      int rem = tix / s;
      it = s - 1;
      for (;; --it) {
        const int cnt = (ng - 1) + (it < g.vlast ? 1 : 0);
        if (rem < cnt) break;
        rem -= cnt;
      }
      grp = rem;
    }
    const int64_t o = (int64_t)(g.grp_lo + grp) * 2 * s * HBO_TILE;
    const int nkb = 8 * s;
    int kb0;
    if (g.mode == 1) { kb0 = 8 * jt; nk = nkb - kb0; csign = 1.f; }
    else { kb0 = 0; nk = 8 * (it + 1); csign = -1.f; }
    cbeta = false;
    ga = g.Xp + (((int64_t)grp * s + it) * nkb + kb0) * NP * P3_CHUNK + tid * 8;
    gb = g.Yp + (((int64_t)grp * s + jt) * nkb + kb0) * NP * P3_CHUNK + tid * 8;
    const int64_t R = o + (int64_t)s * HBO_TILE + (int64_t)it * HBO_TILE, Cc = o + (int64_t)jt * HBO_TILE;
    float* base = static_cast<float*>(g.mode == 1 ? t.S : t.W);
    C = base + R * t.ld + Cc;
  }
  const int wm = wave >> 1, wn = wave & 1;
  const int l32 = lane & 31, lh = lane >> 5;
  auto arr = [&](int st, int op, int p) { return smem + (size_t)((st * 2 + op) * NP + p) * P3_ARR; };

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[a][b][q] = 0.f;

  const int srow = tid >> 1, shalf = tid & 1;
  const int soff = srow * P3_ROW + ((shalf ^ ((srow >> 3) & 1)) * 16);
  struct Slot { u32x4 a0, a1, a2, b0, b1, b2; };
  Slot s0, s1, s2, s3;
#define P3_GLOAD(KT, S)                                                              \
  {                                                                                  \
    const u16* pa_ = ga + (int64_t)(KT) * NP * P3_CHUNK;                              \
    const u16* pb_ = gb + (int64_t)(KT) * NP * P3_CHUNK;                              \
    S.a0 = *reinterpret_cast<const u32x4*>(pa_);                                     \
    S.b0 = *reinterpret_cast<const u32x4*>(pb_);                                     \
    S.a1 = *reinterpret_cast<const u32x4*>(pa_ + P3_CHUNK);                          \
    S.b1 = *reinterpret_cast<const u32x4*>(pb_ + P3_CHUNK);                          \
    if constexpr (!H2) {                                                             \
      S.a2 = *reinterpret_cast<const u32x4*>(pa_ + 2 * P3_CHUNK);                    \
      S.b2 = *reinterpret_cast<const u32x4*>(pb_ + 2 * P3_CHUNK);                    \
    }                                                                                \
  }
#define P3_SSTORE(ST, S)                                                             \
  {                                                                                  \
    *reinterpret_cast<u32x4*>(arr(ST, 0, 0) + soff) = S.a0;                          \
    *reinterpret_cast<u32x4*>(arr(ST, 1, 0) + soff) = S.b0;                          \
    *reinterpret_cast<u32x4*>(arr(ST, 0, 1) + soff) = S.a1;                          \
    *reinterpret_cast<u32x4*>(arr(ST, 1, 1) + soff) = S.b1;                          \
    if constexpr (!H2) {                                                             \
      *reinterpret_cast<u32x4*>(arr(ST, 0, 2) + soff) = S.a2;                        \
      *reinterpret_cast<u32x4*>(arr(ST, 1, 2) + soff) = S.b2;                        \
    }                                                                                \
  }
#define P3_STAGE(KT, CUR, S_FILL, S_NEXT)                                                                         \
  {                                                                                                               \
    const int kt_ = (KT);                                                                                         \
    if (kt_ + 4 < nk) P3_GLOAD(kt_ + 4, S_FILL)                                                                   \
    if constexpr (H2) {                                                                                           \
      f16x8 fa[2][2], fb[2][2];                                                                                   \
      _Pragma("unroll") for (int p = 0; p < 2; ++p)                                                               \
      _Pragma("unroll") for (int tt = 0; tt < 2; ++tt) {                                                          \
        fa[p][tt] = *reinterpret_cast<const f16x8*>(arr(CUR, 0, p) + foff_a + tt * 32 * P3_ROW);                  \
        fb[p][tt] = *reinterpret_cast<const f16x8*>(arr(CUR, 1, p) + foff_b + tt * 32 * P3_ROW);                  \
      }                                                                                                           \
      constexpr int PA[3] = {1, 0, 0};   /* smallest products first: l h', h l', h h' */                          \
      constexpr int PB[3] = {0, 1, 0};                                                                            \
      _Pragma("unroll") for (int q = 0; q < 3; ++q)                                                               \
      _Pragma("unroll") for (int a = 0; a < 2; ++a)                                                               \
      _Pragma("unroll") for (int b = 0; b < 2; ++b)                                                               \
        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[PA[q]][a], fb[PB[q]][b], acc[a][b], 0, 0, 0);       \
    } else {                                                                                                      \
      bf16x8 fa[3][2], fb[3][2];                                                                                  \
      _Pragma("unroll") for (int p = 0; p < 3; ++p)                                                               \
      _Pragma("unroll") for (int tt = 0; tt < 2; ++tt) {                                                          \
        fa[p][tt] = *reinterpret_cast<const bf16x8*>(arr(CUR, 0, p) + foff_a + tt * 32 * P3_ROW);                 \
        fb[p][tt] = *reinterpret_cast<const bf16x8*>(arr(CUR, 1, p) + foff_b + tt * 32 * P3_ROW);                 \
      }                                                                                                           \
      constexpr int PA[6] = {2, 0, 1, 1, 0, 0};   /* smallest products first */                                   \
      constexpr int PB[6] = {0, 2, 1, 0, 1, 0};                                                                   \
      _Pragma("unroll") for (int q = 0; q < 6; ++q)                                                               \
      _Pragma("unroll") for (int a = 0; a < 2; ++a)                                                               \
      _Pragma("unroll") for (int b = 0; b < 2; ++b)                                                               \
        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[PA[q]][a], fb[PB[q]][b], acc[a][b], 0, 0, 0);      \
    }                                                                                                             \
    if (kt_ + 1 < nk) P3_SSTORE((CUR) ^ 1, S_NEXT)                                                                \
    __syncthreads();                                                                                              \
  }
  const int fsw = (lh ^ ((l32 >> 3) & 1)) * 16;
  const int foff_a = (wm * 64 + l32) * P3_ROW + fsw;
  const int foff_b = (wn * 64 + l32) * P3_ROW + fsw;
  P3_GLOAD(0, s0) P3_GLOAD(1, s1) P3_GLOAD(2, s2) P3_GLOAD(3, s3)
  P3_SSTORE(0, s0)
  __syncthreads();
  int ypoll = 0;
  for (int kt0 = 0; kt0 < nk; kt0 += 4) {
    if (yslot) {
      // a panel-chain workgroup is running on this CU: stay off its MFMA / LDS paths until it is done (bounded wait; the table
      // entry is loaded a step ahead, as in gemm.hip)
      if (ypoll != 0)
        for (int spin = 0; spin < 256 && __hip_atomic_load(yslot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0; ++spin)
          __builtin_amdgcn_s_sleep(16);
      ypoll = __hip_atomic_load(yslot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (H2 && augs && kt0 > 0) {
      // the augmented tile-row changes its scale with every chunk of four k blocks and every 16 rows: carry the sums over
      // (row of acc[a][.][q] = 64 wm + 32 a + (q & 3) + 8 (q >> 2) + 4 lh: 16-row group 4 wm + 2 a + (q >> 3))
      const float* sc1 = augs + (int64_t)(kt0 >> 2) * 8 + wm * 4;
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int hq = 0; hq < 2; ++hq) {
          const float f = sc1[a * 2 + hq] / sc1[a * 2 + hq - 8];
#pragma unroll
          for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int q = 0; q < 8; ++q) acc[a][b][hq * 8 + q] *= f;
        }
    }
    P3_STAGE(kt0, 0, s0, s1)
    P3_STAGE(kt0 + 1, 1, s1, s2)
    P3_STAGE(kt0 + 2, 0, s2, s3)
    P3_STAGE(kt0 + 3, 1, s3, s0)
  }
#undef P3_STAGE
#undef P3_GLOAD
#undef P3_SSTORE
  // C = [C] + csign * acc.  Accumulator layout of v_mfma_f32_32x32x16_bf16: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
  float unscale[2][2] = {{1.f, 1.f}, {1.f, 1.f}};   // [a][q >> 3]: H2 -- back from the operands' scales
  if constexpr (H2) {
    const float sy = g.sy_bits ? hbo_h2_scale_for(__uint_as_float(*g.sy_bits)) : g.sy;
    const float sx = g.sx_bits ? hbo_h2_scale_for(__uint_as_float(*g.sx_bits)) : g.sx;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int hq = 0; hq < 2; ++hq)
        unscale[a][hq] = 1.f / ((augs ? augs[(int64_t)((nk - 1) >> 2) * 8 + wm * 4 + a * 2 + hq] : sx) * sy);
  }
  float tmax = 0.f;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      float old[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int row = wm * 64 + a * 32 + (q & 3) + 8 * (q >> 2) + 4 * lh;
        old[q] = cbeta ? gld(C + (int64_t)row * ldc + wn * 64 + b * 32 + l32) : 0.f;
      }
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int row = wm * 64 + a * 32 + (q & 3) + 8 * (q >> 2) + 4 * lh;
        const float v = old[q] + csign * (H2 ? acc[a][b][q] * unscale[a][q >> 3] : acc[a][b][q]);
        gst(C + (int64_t)row * ldc + wn * 64 + b * 32 + l32, v);
        if (H2) tmax = fmaxf(tmax, fabsf(v));
      }
    }
  if (H2 && g.max_out) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) tmax = fmaxf(tmax, __shfl_xor(tmax, o));
    if (lane == 0) atomicMax(g.max_out, __float_as_uint(tmax));
  }
  if (!g.persistent) break;
  }
  if (g.yield_mark) {
    __syncthreads();
    if (tid == 0) yield_leave(g.yield_mark, ytok);
  }
}

}  // namespace

void launch_split3_rows(const float* in, int64_t ld, int row_tiles, unsigned short* out, int nkb, hipStream_t st) {
  if (row_tiles <= 0) return;
  hipLaunchKernelGGL(split3_rows_kernel, dim3((nkb + 3) / 4, row_tiles), dim3(256), 0, st, in, ld, out, nkb);
}
void launch_split3_transpose(const float* in, int64_t ld, int krows, int jcols, unsigned short* out, int nkb, hipStream_t st, int lower_only) {
  if (krows <= 0 || jcols <= 0) return;
  hipLaunchKernelGGL(split3_transpose_kernel<0>, dim3(jcols / 64, krows / 64), dim3(256), 0, st, in, ld, out, nkb, lower_only, (const unsigned int*)nullptr, (unsigned int*)nullptr);
}
// the f16x2 form: one pass for the largest magnitude of what will be split (into *amax_bits, which the caller zeroed -- or that
// already holds a maximum to extend), then the split scaled by the power of two that follows from it
void launch_split2h_transpose_measured(const float* in, int64_t ld, int krows, int jcols, unsigned short* out, int nkb, unsigned int* amax_bits, hipStream_t st, int lower_only) {
  if (krows <= 0 || jcols <= 0) return;
  hipLaunchKernelGGL(split3_transpose_kernel<2>, dim3(jcols / 64, krows / 64), dim3(256), 0, st, in, ld, out, nkb, lower_only, (const unsigned int*)nullptr, amax_bits);
  hipLaunchKernelGGL(split3_transpose_kernel<1>, dim3(jcols / 64, krows / 64), dim3(256), 0, st, in, ld, out, nkb, lower_only, (const unsigned int*)amax_bits, (unsigned int*)nullptr);
}
void launch_post3(const Post3Args& a_in, int col_tiles, hipStream_t st) {
  static unsigned long long seen = 0;
  if (hbo_first_use_on_device(seen))
    hipFuncSetAttribute(reinterpret_cast<const void*>(&post3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, POST3_LDS_BYTES);
  Post3Args a = a_in; a.col_tiles = col_tiles;
  if (a.work_counter) {
    int dev = 0, cus = 256;
    hipGetDevice(&dev); hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const int resident = 2 * cus;   // two workgroups per CU (amdgpu_waves_per_eu(2, 2))
    if (col_tiles * a.nblk > 2 * resident) { hipLaunchKernelGGL(post3_kernel, dim3(resident, 1), dim3(256), POST3_LDS_BYTES, st, a); return; }
    a.work_counter = nullptr;
  }
  hipLaunchKernelGGL(post3_kernel, dim3(col_tiles, a.nblk), dim3(256), POST3_LDS_BYTES, st, a);
}

void launch_split3_panel(const Syrk3Args& a, int row_tiles, int ntasks, hipStream_t st) {
  if (row_tiles <= 0 || a.nk_split <= 0) return;
  if (a.h2) hipLaunchKernelGGL(split3_panel_kernel<true>, dim3((a.nk_split + 3) / 4, row_tiles, ntasks), dim3(256), 0, st, a);
  else hipLaunchKernelGGL(split3_panel_kernel<false>, dim3((a.nk_split + 3) / 4, row_tiles, ntasks), dim3(256), 0, st, a);
}
void launch_syrk3(const Syrk3Args& a, int ntiles, int ntasks, hipStream_t st) {
  if (ntiles <= 0) return;
  static unsigned long long seen = 0;
  if (hbo_first_use_on_device(seen)) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&syrk3_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, POST3_LDS_BYTES + 16);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&syrk3_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, POST3_LDS_BYTES + 16);
  }
  Syrk3Args b = a;
  if (b.persistent <= 0 || !b.work_counter) { b.persistent = 0; b.work_counter = nullptr; }   // no resident grid without a workgroup (small device / knob >= CUs) or a counter
  const int grid = b.persistent > 0 ? std::min(b.persistent, ntiles) : ntiles;
  if (b.h2) hipLaunchKernelGGL(syrk3_kernel<true>, dim3(grid, 1, ntasks), dim3(256), POST3_LDS_BYTES * 2 / 3 + 16, st, b);
  else hipLaunchKernelGGL(syrk3_kernel<false>, dim3(grid, 1, ntasks), dim3(256), POST3_LDS_BYTES + 16, st, b);
}

void launch_split3_block(const Split3Block& a, int ngrp, bool transposed, hipStream_t st) {
  if (a.row_tiles <= 0 || a.nkb <= 0 || ngrp <= 0) return;
  const dim3 gt(a.row_tiles * 2, a.nkb / 4, ngrp), gr((a.nkb + 3) / 4, a.row_tiles, ngrp);   // transposed: 64-column x 64-k tiles
  if (a.h2 && a.max_out) {   // measure first (same elements), then split by the scale that follows from *max_out
    Split3Block b = a; b.scale_bits = a.max_out;
    if (transposed) { hipLaunchKernelGGL(split3_block_t_kernel<2>, gt, dim3(256), 0, st, a); hipLaunchKernelGGL(split3_block_t_kernel<1>, gt, dim3(256), 0, st, b); }
    else { hipLaunchKernelGGL(split3_block_kernel<2>, gr, dim3(256), 0, st, a); hipLaunchKernelGGL(split3_block_kernel<1>, gr, dim3(256), 0, st, b); }
  } else if (a.h2) {
    if (transposed) hipLaunchKernelGGL(split3_block_t_kernel<1>, gt, dim3(256), 0, st, a);
    else hipLaunchKernelGGL(split3_block_kernel<1>, gr, dim3(256), 0, st, a);
  } else {
    if (transposed) hipLaunchKernelGGL(split3_block_t_kernel<0>, gt, dim3(256), 0, st, a);
    else hipLaunchKernelGGL(split3_block_kernel<0>, gr, dim3(256), 0, st, a);
  }
}
