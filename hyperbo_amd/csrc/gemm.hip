// MFMA tile GEMM for gfx950 (CDNA4): the workhorse behind the trailing update of the blocked
// Cholesky (syrk), the recursive triangular inverse (trtri), K^-1 = W^T W (lauum) and the
// posterior V = L^-1 Kxq.  128x128 output tile per 256-thread workgroup (4 waves as 2x2, each
// wave 64x64 = 4x4 v_mfma_{f64,f32}_16x16x4 accumulators), K stepped in 128-byte slabs,
// global -> register -> LDS staging with a two-deep LDS ring (one barrier per K step).
// Every matrix is padded to a multiple of 128 with zeros / identity, so the core has no edge
// predicates; triangular structure is exploited by per-tile K ranges, never by masking.
//
// Replaces, for the reference, what XLA lowers jax.scipy.linalg.cholesky / cho_solve /
// solve_triangular to (hyperbo/basics/linalg.py:29-33,139-145; hyperbo/gp_utils/gp.py:297).
#include "hbo_internal.h"
#include <cstdlib>

namespace {

template <typename T> struct Mma;
template <> struct Mma<double> {
  typedef double acc_t __attribute__((ext_vector_type(4)));
  typedef double vec_t __attribute__((ext_vector_type(2)));
  static __device__ __forceinline__ acc_t mma(double a, double b, acc_t c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  }
  // C/D layout of v_mfma_f64_16x16x4_f64: col = lane&15, row = (lane>>4) + 4*reg
  static __device__ __forceinline__ int crow(int lane, int r) { return (lane >> 4) + 4 * r; }
};
template <> struct Mma<float> {
  typedef float acc_t __attribute__((ext_vector_type(4)));
  typedef float vec_t __attribute__((ext_vector_type(4)));
  static __device__ __forceinline__ acc_t mma(float a, float b, acc_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
  // C/D layout of v_mfma_f32_16x16x4_f32: col = lane&15, row = 4*(lane>>4) + reg
  static __device__ __forceinline__ int crow(int lane, int r) { return 4 * (lane >> 4) + r; }
};

// LDS strides (elements).  k-contiguous operand: [128][SKC]; m-contiguous operand: [BKE][SMC].
//   f64: SKC = 17 (odd stride: the k-contiguous ds_read_b64 fragment reads spread over the banks; the staging stores become
//   pairs of 8-byte writes -- 18, with 16-byte stores and two-way read conflicts, measured equal or slower), f32: SKC = 36.
//   SMC = 144 (== 16 mod 32) for both.  Both layouts take at most 18432 bytes per operand stage.
template <typename T> __device__ __host__ constexpr int skc() { return sizeof(T) == 8 ? 17 : 36; }
template <int TM> __device__ __host__ constexpr int smc() { return TM + 16; }   // == 16 mod 32
constexpr int OPERAND_BYTES = 18432;  // >= 128*skc*sizeof(T) and BKE*SMC*sizeof(T)
constexpr int OPERAND_BYTES_64 = 10240;   // 64-tiles: max(64*skc, BKE*80) elements
constexpr int GEMM_LDS_BYTES_64 = 4 * OPERAND_BYTES_64;
constexpr int GEMM_LDS_BYTES = 4 * OPERAND_BYTES;  // A,B x 2 stages (128-tiles; 64-tiles use half)

template <typename T, bool KC, int TM>
__device__ __forceinline__ void stage_load(const T* __restrict__ g, int64_t ld, int kt,
                                           typename Mma<T>::vec_t (&r)[TM / 32], int tid) {
  typedef typename Mma<T>::vec_t vec_t;
  constexpr int VEC = 16 / sizeof(T);
  constexpr int BKE = 128 / sizeof(T);
  if (KC) {
    const int c = tid & 7, row = tid >> 3;
    const T* p = g + (int64_t)row * ld + (int64_t)kt * BKE + c * VEC;
#pragma unroll
    for (int q = 0; q < TM / 32; ++q) r[q] = gld(reinterpret_cast<const vec_t*>(p + (int64_t)(32 * q) * ld));
  } else {
    constexpr int CPR = TM / VEC;   // 16-byte chunks per TM-element row
    constexpr int RPP = 256 / CPR;  // k rows per pass
    const int c = tid % CPR, kr = tid / CPR;
    const T* p = g + ((int64_t)kt * BKE + kr) * ld + c * VEC;
#pragma unroll
    for (int q = 0; q < TM / 32; ++q) r[q] = gld(reinterpret_cast<const vec_t*>(p + (int64_t)(RPP * q) * ld));
  }
}

template <typename T, bool KC, int TM>
__device__ __forceinline__ void stage_store(T* s, const typename Mma<T>::vec_t (&r)[TM / 32], int tid) {
  typedef typename Mma<T>::vec_t vec_t;
  constexpr int VEC = 16 / sizeof(T);
  if (KC) {
    const int c = tid & 7, row = tid >> 3;
#pragma unroll
    for (int q = 0; q < TM / 32; ++q) {
      if ((skc<T>() * sizeof(T)) % 16 == 0) {
        *reinterpret_cast<vec_t*>(s + (row + 32 * q) * skc<T>() + c * VEC) = r[q];
      } else {
#pragma unroll
        for (int e = 0; e < VEC; ++e) s[(row + 32 * q) * skc<T>() + c * VEC + e] = r[q][e];
      }
    }
  } else {
    constexpr int CPR = TM / VEC;
    constexpr int RPP = 256 / CPR;
    const int c = tid % CPR, kr = tid / CPR;
#pragma unroll
    for (int q = 0; q < TM / 32; ++q) *reinterpret_cast<vec_t*>(s + (kr + RPP * q) * smc<TM>() + c * VEC) = r[q];
  }
}

template <typename T, bool KC, int TM>
__device__ __forceinline__ T frag_read(const T* s, int mn, int k) {
  return KC ? s[mn * skc<T>() + k] : s[k * smc<TM>() + mn];
}

template <typename T>
struct TileJob {
  const T* A; int64_t lda;   // points at (tile row 0, k_begin)
  const T* B; int64_t ldb;   // points at (tile col 0, k_begin)
  T* C; int64_t ldc;         // may be null (POST without V)
  T* colsq;                  // POST: 128 partial column sums of squares, may be null
  int ksteps;
  T alpha;
  int beta;                  // 0: C = alpha*acc ; 1: C += alpha*acc
  int* yield_flag;           // see GemmArgs::yield_flag
};

// (bx, by) = tile coordinates inside a gdx-wide grid: blockIdx of a one-tile-per-workgroup launch, or the tile a
// persistent workgroup drew from the counter
template <typename T, int TM>
__device__ __forceinline__ bool decode_job(const GemmArgs& g, TileJob<T>& j, const int bxi, const int byi, const int gdx, const int task) {
  constexpr int BKE = 128 / sizeof(T);
  const TaskDesc& t = g.tasks[task];
  const int64_t ld = t.ld;
  const int nblk = t.nblk;
  j.colsq = nullptr;
  switch (g.mode) {
    case GEMM_SYRK: {
      // tile coordinates in units of TM (128 or 64); c_lo/c_hi/p0/kt are in 128-units
      constexpr int U = HBO_TILE / TM;
      // plain (r fastest) order.  Measured slower: XCD-aware 8x8 super-tiles that concentrate the 16
      // panels of 64 tiles on one XCD's L2 (61 -> 41 TFLOP/s at N=16384, K=1024).
      // (batches: rows rotated per (column, task) -- in a ragged batch the rows that exist are the low ones of every
      // task and workgroup i runs on XCD (i + const) mod 8)
      const int by = byi;
      const int bx = gridDim.z > 1 ? (int)((bxi + 5 * byi + 3 * task) % gdx) : bxi;
      const int c = g.c_lo * U + by;
      const int r = g.c_lo * U + bx;
      const int nrt = (nblk + ((g.aug & 1) ? 1 : 0)) * U;
      const int chi = (g.c_hi < nblk ? g.c_hi : nblk) * U;
      if (c >= chi || r < c || r >= nrt) return false;
      T* Am = static_cast<T*>(t.A);
      j.A = Am + (int64_t)r * TM * ld + (int64_t)g.p0 * HBO_TILE;
      j.B = Am + (int64_t)c * TM * ld + (int64_t)g.p0 * HBO_TILE;
      j.C = Am + (int64_t)r * TM * ld + (int64_t)c * TM;
      j.lda = j.ldb = j.ldc = ld;
      j.ksteps = g.kt * HBO_TILE / BKE;
      j.alpha = (T)-1; j.beta = 1;
#ifdef HBO_GEMM_DEBUG
      if (g.aug & 2) j.beta = 0;
      if (g.aug & 4) { j.C = nullptr; j.beta = 0; }
#endif
      return true;
    }
    case GEMM_TRTRI_A:
    case GEMM_TRTRI_B: {
      // Tile coordinates in units of TM (128, or 64 for small / batched problems); s is in 128-units.
      // blockIdx.y carries the index that fixes the K length, ordered longest first, so that the
      // dispatcher hands out tiles in longest-processing-time order (x runs over groups x tiles).
      constexpr int U = HBO_TILE / TM;
      const int s = g.p0, su = s * U;
      const int grp = g.grp_lo + bxi / su;
      const int inner = bxi % su;
      int jt, it;
      if (g.mode == GEMM_TRTRI_A) {
        // K = s*128 - jt*TM.  Ragged batches: the row tiles that exist are the low ones in every task and workgroup i
        // runs on XCD (i + const) mod 8, so the row order is rotated per (column, task)
        jt = byi;
        const int vu = (grp == g.c_hi ? g.c_lo : s) * U;   // tile rows launched for this group (the last may be cut)
        it = (inner + 5 * byi + 3 * task) % vu;
      }
      else { it = (g.kt > 0 ? g.kt * U : su) - 1 - byi; jt = inner; }   // K = (it+1)*TM; kt = valid rows (single group)
      const int64_t o = (int64_t)grp * 2 * s * HBO_TILE;                      // element offsets from here on
      const int64_t R = o + (int64_t)s * HBO_TILE + (int64_t)it * TM;
      if (R >= (int64_t)nblk * HBO_TILE) return false;
      const int64_t Cc = o + (int64_t)jt * TM;
      const T* L = static_cast<const T*>(t.A);
      T* W = static_cast<T*>(t.W);
      T* S = static_cast<T*>(t.S);
      if (g.mode == GEMM_TRTRI_A) {
        // S21 = L21 * W11 ; W11 lower-triangular => k >= column tile start
        j.A = L + R * ld + Cc;
        j.B = W + Cc * ld + Cc;
        j.C = S + R * ld + Cc;
        j.ksteps = (s * HBO_TILE - jt * TM) / BKE;
        j.alpha = (T)1;
      } else {
        // W21 = -W22 * S21 ; W22 lower-triangular => k < row tile end
        const int64_t o2 = o + (int64_t)s * HBO_TILE;
        j.A = W + R * ld + o2;
        j.B = S + o2 * ld + Cc;
        j.C = W + R * ld + Cc;
        j.ksteps = (it + 1) * TM / BKE;
        j.alpha = (T)-1;
      }
      j.lda = j.ldb = j.ldc = ld;
      j.beta = 0;
      return true;
    }
    case GEMM_LAUUM: {
      // C[i,j] = sum_{k >= i*TM} W[k, i-tile]^T W[k, j-tile]   (rows of W[:,j] above j*TM are zero), lower tiles.
      constexpr int U = HBO_TILE / TM;
      int i, jt;
      if (U == 1) {
        // 128-tiles: 1-D grid over the n(n+1)/2 real tiles only (no workgroups that exit at once), row tile i slow =
        // K = (n - i) blocks descending (longest first), column tile jt fast.  Consecutive workgroups -- which the
        // dispatcher deals to the 8 XCDs round-robin -- then have equal K and share the row panel W[:, i].
        // History (in-kernel stamps, tools/gemm_wall.py): 2-D grid with i fast: 370 of 512 slots busy, 3.56 ms (XCD 0
        // always received the longest of every 8 tiles); column-major with every second group of 8 reversed: 412
        // slots, 3.38 ms; this order: 481 slots, 3.08 ms.
        int lin = bxi;
        if (lin >= nblk * (nblk + 1) / 2) return false;
        i = 0;
        while (lin > i) { lin -= i + 1; ++i; }
        jt = lin;
      } else {
        // 64-tiles (small / batched matrices): same row-major order; the tile right of an even diagonal tile is
        // computed too, so that every 128x128 block on the diagonal is complete (the contraction kernel reads whole
        // 128-blocks): row i has (i | 1) + 1 tiles, 2 n (n + 1) in total
        int lin = bxi;
        if (lin >= 2 * nblk * (nblk + 1)) return false;
        i = 0;
        while (lin > (i | 1)) { lin -= (i | 1) + 1; ++i; }
        jt = lin;
      }
      const int64_t k0 = (int64_t)i * TM;
      const T* W = static_cast<const T*>(t.W);
      j.A = W + k0 * ld + (int64_t)i * TM;
      j.B = W + k0 * ld + (int64_t)jt * TM;
      j.C = static_cast<T*>(t.S) + (int64_t)i * TM * ld + (int64_t)jt * TM;
      j.lda = j.ldb = j.ldc = ld;
      j.ksteps = (int)(((int64_t)nblk * HBO_TILE - k0) / BKE);
      j.alpha = (T)1; j.beta = 0;
      return true;
    }
    case GEMM_SWEEP_B: {
      // W[it, jt] = -sum_{k = b0 .. it} W[it, k] T[k, jt]: rows of the group (longest K = lowest row first), columns left of it
      constexpr int U = HBO_TILE / TM;
      const int b0 = g.c_lo, b1 = g.c_hi < nblk ? g.c_hi : nblk;
      const int rows = (b1 - b0) * U;
      if (byi >= rows || bxi >= b0 * U) return false;
      const int it = b0 * U + rows - 1 - byi, jt = bxi;
      const int64_t k0 = (int64_t)b0 * HBO_TILE;
      T* W = static_cast<T*>(t.W);
      j.A = W + (int64_t)it * TM * ld + k0;
      j.B = static_cast<const T*>(t.S) + k0 * ld + (int64_t)jt * TM;
      j.C = W + (int64_t)it * TM * ld + (int64_t)jt * TM;
      j.lda = j.ldb = j.ldc = ld;
      j.ksteps = (int)(((int64_t)(it + 1) * TM - k0) / BKE);
      j.alpha = (T)-1; j.beta = 0;
      return true;
    }
    case GEMM_SWEEP_T: {
      // T[it, jt] (+)= sum_{k = max(jt, b0) .. b1} L[it, k] W[k, jt]: rows below the group (rotated per column and task in a
      // batch, see SYRK), columns up to its end -- those left of the group (full K) first
      constexpr int U = HBO_TILE / TM;
      const int b0 = g.c_lo, b1 = g.c_hi;
      if (b1 >= nblk) return false;                       // nothing below the front in this task
      const int below = (nblk - b1) * U;
      const int bx = gridDim.z > 1 || g.ptasks > 1 ? (int)((bxi + 5 * byi + 3 * task) % gdx) : bxi;
      if (bx >= below || byi >= b1 * U) return false;
      const int it = b1 * U + bx, jt = byi;
      const int64_t kb = (int64_t)b0 * HBO_TILE, kj = (int64_t)jt * TM;
      const int64_t k0 = kj > kb ? kj : kb;
      j.A = static_cast<const T*>(t.A) + (int64_t)it * TM * ld + k0;
      j.B = static_cast<const T*>(t.W) + k0 * ld + kj;
      j.C = static_cast<T*>(t.S) + (int64_t)it * TM * ld + kj;
      j.lda = j.ldb = j.ldc = ld;
      j.ksteps = (int)(((int64_t)b1 * HBO_TILE - k0) / BKE);
      j.alpha = (T)1; j.beta = kj >= kb ? 0 : 1;
      return true;
    }
    case GEMM_SWEEP_C: {
      // K^-1[i, jt] (+)= sum_{k = max(i, b0) .. b1} W[k, i]^T W[k, jt] over the lower tiles of the leading b1 blocks, enumerated as
      // LAUUM does (row tile i slow; 64-tiles: the tile right of an even diagonal tile too, so that every diagonal 128-block is whole)
      constexpr int U = HBO_TILE / TM;
      const int b0 = g.c_lo, b1 = g.c_hi < nblk ? g.c_hi : nblk;
      if (b1 <= b0) return false;                         // this task ended before the group
      int i = 0, lin = bxi, jt;
      if (U == 1) {
        if (lin >= b1 * (b1 + 1) / 2) return false;
        while (lin > i) { lin -= i + 1; ++i; }
      } else {
        if (lin >= 2 * b1 * (b1 + 1)) return false;
        while (lin > (i | 1)) { lin -= (i | 1) + 1; ++i; }
      }
      jt = lin;
      const int64_t kb = (int64_t)b0 * HBO_TILE, ki = (int64_t)i * TM;
      const int64_t k0 = ki > kb ? ki : kb;
      const T* W = static_cast<const T*>(t.W);
      j.A = W + k0 * ld + ki;
      j.B = W + k0 * ld + (int64_t)jt * TM;
      j.C = static_cast<T*>(t.S) + ki * ld + (int64_t)jt * TM;
      j.lda = j.ldb = j.ldc = ld;
      j.ksteps = (int)(((int64_t)b1 * HBO_TILE - k0) / BKE);
      j.alpha = (T)1; j.beta = ki >= kb ? 0 : 1;
      return true;
    }
    case GEMM_VTV: {
      // C[i, j] -= sum_k V[k, i-tile]^T V[k, j-tile]: V = g.B (npad x ldb, candidates contiguous), C = g.V (same leading dimension)
      const T* Vm = static_cast<const T*>(g.B);
      j.A = Vm + (int64_t)byi * HBO_TILE;
      j.B = Vm + (int64_t)bxi * HBO_TILE;
      j.C = static_cast<T*>(g.V) + (int64_t)byi * HBO_TILE * g.ldb + (int64_t)bxi * HBO_TILE;
      j.lda = j.ldb = j.ldc = g.ldb;
      j.ksteps = nblk * HBO_TILE / BKE;
      j.alpha = (T)-1; j.beta = 1;
      return true;
    }
    case GEMM_POST: {
      if (g.kchunk > 0) {
        // split-K form: byi enumerates (row tile i, chunk ch), ch < ceil((i + 1) / kchunk)
        int i = 0, rem = byi;
        for (;; ++i) {
          if (i >= nblk) return false;
          const int nch = (i + g.kchunk) / g.kchunk;
          if (rem < nch) break;
          rem -= nch;
        }
        const int k0 = rem * g.kchunk;                       // first K block of the chunk
        const int kb = (i + 1 - k0 < g.kchunk ? i + 1 - k0 : g.kchunk);
        const T* W = static_cast<const T*>(t.W);
        j.A = W + (int64_t)i * HBO_TILE * ld + (int64_t)k0 * HBO_TILE;
        j.lda = ld;
        j.B = static_cast<const T*>(g.B) + (int64_t)k0 * HBO_TILE * g.ldb + (int64_t)bxi * HBO_TILE;
        j.ldb = g.ldb;
        j.C = static_cast<T*>(g.V) + ((int64_t)rem * nblk * HBO_TILE + (int64_t)i * HBO_TILE) * g.ldb + (int64_t)bxi * HBO_TILE;
        j.ldc = g.ldb;
        j.ksteps = kb * HBO_TILE / BKE;
        j.alpha = (T)1; j.beta = 0;
        return true;
      }
      const int i = nblk - 1 - byi;  // heavy (long K) row tiles first
      if (i < 0) return false;
      const int jq = bxi;
      const T* W = static_cast<const T*>(t.W);
      j.A = W + (int64_t)i * HBO_TILE * ld;
      j.lda = ld;
      j.B = static_cast<const T*>(g.B) + (int64_t)jq * HBO_TILE;
      j.ldb = g.ldb;
      j.C = g.V ? static_cast<T*>(g.V) + (int64_t)i * HBO_TILE * g.ldb + (int64_t)jq * HBO_TILE : nullptr;
      j.ldc = g.ldb;
      j.colsq = g.colsq ? static_cast<T*>(g.colsq) + (int64_t)i * g.ldb + (int64_t)jq * HBO_TILE : nullptr;
      j.ksteps = (i + 1) * HBO_TILE / BKE;
      j.alpha = (T)1; j.beta = 0;
      return true;
    }
  }
  return false;
}

template <typename T, bool AKC, bool BKC, int TM>
__device__ __forceinline__ void gemm_tile(const TileJob<T>& job, unsigned char* smem) {
  typedef typename Mma<T>::acc_t acc_t;
  typedef typename Mma<T>::vec_t vec_t;
  constexpr int BKE = 128 / sizeof(T);
  constexpr int MI = TM / 32;            // 16x16 MFMA tiles per wave and dimension (wave tile TM/2)
  constexpr int WT = TM / 2;             // wave tile edge
  constexpr int OPB = TM == 128 ? OPERAND_BYTES : OPERAND_BYTES_64;

  T* sA0 = reinterpret_cast<T*>(smem);
  T* sA1 = reinterpret_cast<T*>(smem + OPB);
  T* sB0 = reinterpret_cast<T*>(smem + 2 * OPB);
  T* sB1 = reinterpret_cast<T*>(smem + 3 * OPB);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l15 = lane & 15, lq = lane >> 4;

  // accumulators start from C/alpha when the tile is accumulated into C (beta = 1): the C tile is
  // fetched together with the first operand slabs instead of in a read-modify-write epilogue
  acc_t acc[MI][MI];
  if (job.beta) {
    const T inv_alpha = (T)1 / job.alpha;
#pragma unroll
    for (int a = 0; a < MI; ++a)
#pragma unroll
      for (int b = 0; b < MI; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = wm * WT + a * 16 + Mma<T>::crow(lane, r);
          const int col = wn * WT + b * 16 + l15;
          acc[a][b][r] = gld(job.C + (int64_t)row * job.ldc + col) * inv_alpha;
        }
  } else {
#pragma unroll
    for (int a = 0; a < MI; ++a)
#pragma unroll
      for (int b = 0; b < MI; ++b) acc[a][b] = (acc_t){0, 0, 0, 0};
  }

  const int nk = job.ksteps;
  vec_t ra[MI], rb[MI];
  stage_load<T, AKC, TM>(job.A, job.lda, 0, ra, tid);
  stage_load<T, BKC, TM>(job.B, job.ldb, 0, rb, tid);
  stage_store<T, AKC, TM>(sA0, ra, tid);
  stage_store<T, BKC, TM>(sB0, rb, tid);
  __syncthreads();

  // Yield poll, software-pipelined: the table entry of this CU is loaded at the top of a K step -- ahead of the slab
  // prefetch, so it has arrived, at no cost, by the time the prefetch is waited for -- and looked at at the top of the
  // NEXT step (a poll that is waited for on the spot costs an L2 round trip per K step: N = 16384 lost 0.9 % to it).
  const int* const yslot = job.yield_flag ? job.yield_flag + cu_token() : nullptr;
  int ypoll = 0;
  for (int kt = 0; kt < nk; ++kt) {
    if (yslot) {
      // a panel-chain workgroup is running on this CU: stay off its MFMA / LDS paths until it is done (bounded wait)
      if (ypoll != 0)
        for (int spin = 0; spin < 256 && __hip_atomic_load(yslot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0; ++spin)
          __builtin_amdgcn_s_sleep(16);
      ypoll = __hip_atomic_load(yslot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const T* cA = (kt & 1) ? sA1 : sA0;
    const T* cB = (kt & 1) ? sB1 : sB0;
    const bool more = (kt + 1 < nk);
    if (more) {
      stage_load<T, AKC, TM>(job.A, job.lda, kt + 1, ra, tid);
      stage_load<T, BKC, TM>(job.B, job.ldb, kt + 1, rb, tid);
    }
#pragma unroll
    for (int kk = 0; kk < BKE / 4; ++kk) {
      const int k = kk * 4 + lq;
      T af[MI], bf[MI];
#pragma unroll
      for (int a = 0; a < MI; ++a) af[a] = frag_read<T, AKC, TM>(cA, wm * WT + a * 16 + l15, k);
#pragma unroll
      for (int b = 0; b < MI; ++b) bf[b] = frag_read<T, BKC, TM>(cB, wn * WT + b * 16 + l15, k);
#pragma unroll
      for (int a = 0; a < MI; ++a)
#pragma unroll
        for (int b = 0; b < MI; ++b) acc[a][b] = Mma<T>::mma(af[a], bf[b], acc[a][b]);
      // the LDS stores of the prefetched slab go in the middle of the MFMA stream
      if (kk == BKE / 4 - 2 && more) {
        __builtin_amdgcn_sched_barrier(0);
        stage_store<T, AKC, TM>((kt & 1) ? sA0 : sA1, ra, tid);
        stage_store<T, BKC, TM>((kt & 1) ? sB0 : sB1, rb, tid);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __syncthreads();
  }

  // epilogue: C = alpha * acc (the old C, if any, is already inside acc)
  if (job.C) {
#pragma unroll
    for (int a = 0; a < MI; ++a)
#pragma unroll
      for (int b = 0; b < MI; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = wm * WT + a * 16 + Mma<T>::crow(lane, r);
          const int col = wn * WT + b * 16 + l15;
          gst(job.C + (int64_t)row * job.ldc + col, job.alpha * acc[a][b][r]);
        }
  }
  if (TM == 128 && job.colsq) {
    // sum over this tile's 128 rows of acc^2, per column
    T* red = reinterpret_cast<T*>(smem);  // [4 waves][64]
    T part[MI];
#pragma unroll
    for (int b = 0; b < MI; ++b) {
      T s = 0;
#pragma unroll
      for (int a = 0; a < MI; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) s += acc[a][b][r] * acc[a][b][r];
      s += __shfl_xor(s, 16);
      s += __shfl_xor(s, 32);
      part[b] = s;
    }
    // (the k-loop ended with a barrier, so smem is free)
    if (lq == 0) {
#pragma unroll
      for (int b = 0; b < MI; ++b) red[wave * 64 + b * 16 + l15] = part[b];
    }
    __syncthreads();
    if (tid < 128) {
      const int wn2 = tid >> 6, c = tid & 63;
      // waves (wm=0,wn2) and (wm=1,wn2)
      gst(job.colsq + wn2 * 64 + c, red[(0 * 2 + wn2) * 64 + c] + red[(1 * 2 + wn2) * 64 + c]);
    }
  }
}

// ---- the pipelined tile core (round 4) ------------------------------------------------------------------------------------------
// Same tile, same LDS layouts, same arithmetic (every accumulator receives its products in ascending k: bit-identical results) --
// a different pipeline.  gemm_tile issues the LDS fragment reads of a k step right before the 16 MFMAs that consume them, stores a
// whole slab to LDS in one run of instructions and keeps two LDS stages per operand; in its K loop a wave has stretches of 40-60
// instructions without an MFMA, and it reaches 0.80-0.85 of the fp64 MFMA peak where a vendor kernel of the same macro tile
// (rocBLAS dgemm, 128 x 128 x 16, 256 threads) sustains the peak (profiles/r04_gemm_pipeline.md).  What this core does instead:
//   * the fragments of k step i + 1 are read (into a second register set) between the first MFMAs of step i;
//   * ONE LDS stage per operand and two barriers per slab: after the last fragment read of a slab every wave passes a barrier,
//     the next slab is written from the prefetch registers, a second barrier, then its first fragments are read -- every one of
//     those instructions between two MFMAs of the slab's last two k steps (18 MFMAs between the barriers, 5 behind the second);
//   * global prefetch two slabs deep: each LDS write is followed, one MFMA later, by the buffer load that refills its registers
//     (scalar base that moves with the slab + one 32-bit offset register per load);
//   * the order is pinned with scheduling barriers: the compiler's own order gathers the LDS and memory instructions again.
// 36 KB of LDS instead of 72.
template <typename T>
__device__ __forceinline__ unsigned long long uniform_addr(const T* p) {
  const unsigned long long v = reinterpret_cast<unsigned long long>(p);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t tile_rsrc(unsigned long long addr) {
  return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(addr), 0, 0x7fffffff, 0x00020000);
}
template <typename T, bool KC, int TM>
__device__ __forceinline__ void stage_offsets(int64_t ld, int (&off)[TM / 32], int tid) {   // byte offsets inside one K slab
  constexpr int VEC = 16 / sizeof(T);
  if (KC) {
    const int c = tid & 7, row = tid >> 3;
#pragma unroll
    for (int q = 0; q < TM / 32; ++q) off[q] = (int)(((int64_t)(row + 32 * q) * ld + c * VEC) * (int64_t)sizeof(T));
  } else {
    constexpr int CPR = TM / VEC, RPP = 256 / CPR;
    const int c = tid % CPR, kr = tid / CPR;
#pragma unroll
    for (int q = 0; q < TM / 32; ++q) off[q] = (int)(((int64_t)(kr + RPP * q) * ld + c * VEC) * (int64_t)sizeof(T));
  }
}
// one 16-byte piece of a slab: global -> registers, registers -> LDS
template <typename T>
__device__ __forceinline__ typename Mma<T>::vec_t piece_load(unsigned long long slab, int off) {
  return __builtin_bit_cast(typename Mma<T>::vec_t, __builtin_amdgcn_raw_buffer_load_b128(tile_rsrc(slab), off, 0, 0));
}
template <typename T, bool KC, int TM>
__device__ __forceinline__ void piece_store(T* s, const typename Mma<T>::vec_t& r, int q, int tid) {
  typedef typename Mma<T>::vec_t vec_t;
  constexpr int VEC = 16 / sizeof(T);
  if (KC) {
    const int c = tid & 7, row = tid >> 3;
    if ((skc<T>() * sizeof(T)) % 16 == 0) {
      *reinterpret_cast<vec_t*>(s + (row + 32 * q) * skc<T>() + c * VEC) = r;
    } else {
#pragma unroll
      for (int e = 0; e < VEC; ++e) s[(row + 32 * q) * skc<T>() + c * VEC + e] = r[e];
    }
  } else {
    constexpr int CPR = TM / VEC, RPP = 256 / CPR;
    const int c = tid % CPR, kr = tid / CPR;
    *reinterpret_cast<vec_t*>(s + (kr + RPP * q) * smc<TM>() + c * VEC) = r;
  }
}

template <typename T, bool AKC, bool BKC, int TM>
__device__ __forceinline__ void gemm_tile2(const TileJob<T>& job, unsigned char* smem) {
  typedef typename Mma<T>::acc_t acc_t;
  typedef typename Mma<T>::vec_t vec_t;
  constexpr int BKE = 128 / sizeof(T);
  constexpr int KK = BKE / 4;            // MFMA k steps per slab
  constexpr int MI = TM / 32;
  constexpr int WT = TM / 2;
  static_assert(MI == 4 && KK >= 4 && KK % 2 == 0, "the interleave below is written for the 128-tile: 16 MFMAs and 8 fragments per k step, 8 pieces per slab");
  T* sA = reinterpret_cast<T*>(smem);
  T* sB = reinterpret_cast<T*>(smem + OPERAND_BYTES);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l15 = lane & 15, lq = lane >> 4;

  acc_t acc[MI][MI];
  if (job.beta) {
    const T inv_alpha = (T)1 / job.alpha;
#pragma unroll
    for (int a = 0; a < MI; ++a)
#pragma unroll
      for (int b = 0; b < MI; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = wm * WT + a * 16 + Mma<T>::crow(lane, r);
          const int col = wn * WT + b * 16 + l15;
          acc[a][b][r] = gld(job.C + (int64_t)row * job.ldc + col) * inv_alpha;
        }
  } else {
#pragma unroll
    for (int a = 0; a < MI; ++a)
#pragma unroll
      for (int b = 0; b < MI; ++b) acc[a][b] = (acc_t){0, 0, 0, 0};
  }

  const int nk = job.ksteps;
  int offa[MI], offb[MI];
  stage_offsets<T, AKC, TM>(job.lda, offa, tid);
  stage_offsets<T, BKC, TM>(job.ldb, offb, tid);
  // the slab's origin moves with the (uniform) base address: no 32-bit offset ever spans more than one slab of one tile
  unsigned long long slabA = uniform_addr(job.A), slabB = uniform_addr(job.B);
  const unsigned long long stepA = uniform_addr(reinterpret_cast<const char*>((AKC ? (int64_t)BKE : (int64_t)BKE * job.lda) * (int64_t)sizeof(T)));
  const unsigned long long stepB = uniform_addr(reinterpret_cast<const char*>((BKC ? (int64_t)BKE : (int64_t)BKE * job.ldb) * (int64_t)sizeof(T)));
  vec_t ra[MI], rb[MI];
#pragma unroll
  for (int q = 0; q < MI; ++q) { ra[q] = piece_load<T>(slabA, offa[q]); rb[q] = piece_load<T>(slabB, offb[q]); }
#pragma unroll
  for (int q = 0; q < MI; ++q) { piece_store<T, AKC, TM>(sA, ra[q], q, tid); piece_store<T, BKC, TM>(sB, rb[q], q, tid); }
  if (nk > 1) { slabA += stepA; slabB += stepB; }
#pragma unroll
  for (int q = 0; q < MI; ++q) { ra[q] = piece_load<T>(slabA, offa[q]); rb[q] = piece_load<T>(slabB, offb[q]); }
  __syncthreads();

  T af[2][MI], bf[2][MI];
#define HBO_SB() __builtin_amdgcn_sched_barrier(0)
  // fragment i of k step kk (0-3: A, 4-7: B) into register set `set`
  auto frag = [&](int set, int kk, int i) {
    const int k = kk * 4 + lq;
    if (i < MI) af[set][i] = frag_read<T, AKC, TM>(sA, wm * WT + i * 16 + l15, k);
    else bf[set][i - MI] = frag_read<T, BKC, TM>(sB, wn * WT + (i - MI) * 16 + l15, k);
  };
  // MFMA i of a k step, serpentine over the 4 x 4 accumulators: consecutive MFMAs share an operand register
  auto mma = [&](int set, int i) {
    const int a = i / MI, b = (a & 1) ? MI - 1 - i % MI : i % MI;
    acc[a][b] = Mma<T>::mma(af[set][a], bf[set][b], acc[a][b]);
  };
  // a k step whose only company is the next step's fragment reads: two behind each of the first four MFMAs
  auto step_reads = [&](int set, int nset, int nkk) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      mma(set, i);
      if (i < 4) { HBO_SB(); frag(nset, nkk, 2 * i); frag(nset, nkk, 2 * i + 1); HBO_SB(); }
    }
  };
  // piece j (0-3: A, 4-7: B) of the prefetched slab to LDS / refilled from the slab after it
  auto piece_out = [&](int j) {
    if (j < MI) piece_store<T, AKC, TM>(sA, ra[j], j, tid); else piece_store<T, BKC, TM>(sB, rb[j - MI], j - MI, tid);
  };
  auto piece_in = [&](int j) {
    if (j < MI) ra[j] = piece_load<T>(slabA, offa[j]); else rb[j - MI] = piece_load<T>(slabB, offb[j - MI]);
  };
#pragma unroll
  for (int i = 0; i < 2 * MI; ++i) frag(0, 0, i);

  const int* const yslot = job.yield_flag ? job.yield_flag + cu_token() : nullptr;
  int ypoll = 0;
  for (int kt = 0; kt + 1 < nk; ++kt) {
    if (yslot) {
      // a panel-chain workgroup is running on this CU: stay off its MFMA / LDS paths until it is done (bounded wait; see gemm_tile)
      if (ypoll != 0)
        for (int spin = 0; spin < 256 && __hip_atomic_load(yslot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0; ++spin)
          __builtin_amdgcn_s_sleep(16);
      ypoll = __hip_atomic_load(yslot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // the registers hold slab kt + 1; what refills them is slab kt + 2 (the last slab again when there is none: no branch in here)
    if (kt + 2 < nk) { slabA += stepA; slabB += stepB; }
#pragma unroll
    for (int kk = 0; kk < KK - 2; ++kk) step_reads(kk & 1, (kk + 1) & 1, kk + 1);
    // the last k step but one: the last fragments of the slab, the first barrier, then the first four pieces
#pragma unroll
    for (int i = 0; i < 7; ++i) {
      mma(0, i);
      if (i < 4) { HBO_SB(); frag(1, KK - 1, 2 * i); frag(1, KK - 1, 2 * i + 1); HBO_SB(); }
    }
    HBO_SB(); __syncthreads(); HBO_SB();                      // every wave has read the slab's last fragments
    mma(0, 7);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      HBO_SB(); piece_out(j); HBO_SB(); mma(0, 8 + 2 * j); HBO_SB(); piece_in(j); HBO_SB(); mma(0, 9 + 2 * j);
    }
    // the last k step: the other four pieces, the second barrier, the next slab's first fragments
#pragma unroll
    for (int j = 4; j < 8; ++j) {
      HBO_SB(); piece_out(j); HBO_SB(); mma(1, 2 * (j - 4)); HBO_SB(); piece_in(j); HBO_SB(); mma(1, 2 * (j - 4) + 1);
    }
    mma(1, 8); mma(1, 9);
    HBO_SB(); __syncthreads(); HBO_SB();                      // the new slab is in LDS
    mma(1, 10);
#pragma unroll
    for (int i = 0; i < 4; ++i) { HBO_SB(); frag(0, 0, 2 * i); frag(0, 0, 2 * i + 1); HBO_SB(); mma(1, 11 + i); }
    mma(1, 15);
    HBO_SB();
  }
  // the last slab
#pragma unroll
  for (int kk = 0; kk < KK - 1; ++kk) step_reads(kk & 1, (kk + 1) & 1, kk + 1);
#pragma unroll
  for (int i = 0; i < 16; ++i) mma(1, i);
#undef HBO_SB
  if (yslot) __syncthreads();   // (uniform exit of the K loop for the LDS reuse of the epilogue below)

  if (job.C) {
#pragma unroll
    for (int a = 0; a < MI; ++a)
#pragma unroll
      for (int b = 0; b < MI; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = wm * WT + a * 16 + Mma<T>::crow(lane, r);
          const int col = wn * WT + b * 16 + l15;
          gst(job.C + (int64_t)row * job.ldc + col, job.alpha * acc[a][b][r]);
        }
  }
  if (TM == 128 && job.colsq) {
    __syncthreads();   // every wave is done with the last slab's fragments: the LDS is free for the reduction
    T* red = reinterpret_cast<T*>(smem);  // [4 waves][64]
    T part[MI];
#pragma unroll
    for (int b = 0; b < MI; ++b) {
      T s_ = 0;
#pragma unroll
      for (int a = 0; a < MI; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) s_ += acc[a][b][r] * acc[a][b][r];
      s_ += __shfl_xor(s_, 16);
      s_ += __shfl_xor(s_, 32);
      part[b] = s_;
    }
    if (lq == 0) {
#pragma unroll
      for (int b = 0; b < MI; ++b) red[wave * 64 + b * 16 + l15] = part[b];
    }
    __syncthreads();
    if (tid < 128) {
      const int wn2 = tid >> 6, c = tid & 63;
      gst(job.colsq + wn2 * 64 + c, red[(0 * 2 + wn2) * 64 + c] + red[(1 * 2 + wn2) * 64 + c]);
    }
  }
}
// ---- the 64-tile core (round 4) ------------------------------------------------------------------------------------------------
// The single-stage scheme of gemm_tile2 for the 64 x 64 tile: 4 MFMAs and 4 fragment reads per wave and k step, two barriers per
// slab, fragment reads of the next k step beside the MFMAs of the current one, and a ring of PF slabs prefetched into registers
// (16 VGPRs per slab and thread).  PF = 2: a lone workgroup's slabs come from L2 at ~1 us each, a deeper ring buys nothing and
// costs the fourth workgroup per CU (profiles/r04_gemm_pipeline.md, section 4).  Same ascending-k arithmetic: identical results.
#ifndef HBO_PF64
#define HBO_PF64 2
#endif
#ifndef HBO_LB64
#define HBO_LB64 4
#endif
template <typename T, bool AKC, bool BKC, int TM, int PF>
__device__ __forceinline__ void gemm_tile3(const TileJob<T>& job, unsigned char* smem) {
  typedef typename Mma<T>::acc_t acc_t;
  typedef typename Mma<T>::vec_t vec_t;
  constexpr int BKE = 128 / sizeof(T);
  constexpr int KK = BKE / 4;
  constexpr int MI = TM / 32;
  constexpr int WT = TM / 2;
  static_assert(MI == 2 && KK >= 4 && KK % 2 == 0, "written for the 64-tile: 4 MFMAs and 4 fragments per k step, 4 pieces per slab");
  T* sA = reinterpret_cast<T*>(smem);
  T* sB = reinterpret_cast<T*>(smem + OPERAND_BYTES_64);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l15 = lane & 15, lq = lane >> 4;

  acc_t acc[MI][MI];
  if (job.beta) {
    const T inv_alpha = (T)1 / job.alpha;
#pragma unroll
    for (int a = 0; a < MI; ++a)
#pragma unroll
      for (int b = 0; b < MI; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = wm * WT + a * 16 + Mma<T>::crow(lane, r);
          const int col = wn * WT + b * 16 + l15;
          acc[a][b][r] = gld(job.C + (int64_t)row * job.ldc + col) * inv_alpha;
        }
  } else {
#pragma unroll
    for (int a = 0; a < MI; ++a)
#pragma unroll
      for (int b = 0; b < MI; ++b) acc[a][b] = (acc_t){0, 0, 0, 0};
  }

  const int nk = job.ksteps;
  int offa[MI], offb[MI];
  stage_offsets<T, AKC, TM>(job.lda, offa, tid);
  stage_offsets<T, BKC, TM>(job.ldb, offb, tid);
  unsigned long long slabA = uniform_addr(job.A), slabB = uniform_addr(job.B);   // origin of the next slab to be fetched
  const unsigned long long stepA = uniform_addr(reinterpret_cast<const char*>((AKC ? (int64_t)BKE : (int64_t)BKE * job.lda) * (int64_t)sizeof(T)));
  const unsigned long long stepB = uniform_addr(reinterpret_cast<const char*>((BKC ? (int64_t)BKE : (int64_t)BKE * job.ldb) * (int64_t)sizeof(T)));
  vec_t ra[PF][MI], rb[PF][MI];
  auto fetch = [&](int set) {      // the next slab into register set `set`
#pragma unroll
    for (int q = 0; q < MI; ++q) { ra[set][q] = piece_load<T>(slabA, offa[q]); rb[set][q] = piece_load<T>(slabB, offb[q]); }
    slabA += stepA; slabB += stepB;
  };
  auto to_lds = [&](int set) {
#pragma unroll
    for (int q = 0; q < MI; ++q) { piece_store<T, AKC, TM>(sA, ra[set][q], q, tid); piece_store<T, BKC, TM>(sB, rb[set][q], q, tid); }
  };
#pragma unroll
  for (int u = 0; u < PF; ++u)
    if (u < nk) fetch(u);
  to_lds(0);
  if (PF < nk) fetch(0);
  __syncthreads();

  T af[2][MI], bf[2][MI];
#define HBO_SB() __builtin_amdgcn_sched_barrier(0)
  auto frags = [&](int set, int kk) {
    const int k = kk * 4 + lq;
#pragma unroll
    for (int a = 0; a < MI; ++a) af[set][a] = frag_read<T, AKC, TM>(sA, wm * WT + a * 16 + l15, k);
#pragma unroll
    for (int b = 0; b < MI; ++b) bf[set][b] = frag_read<T, BKC, TM>(sB, wn * WT + b * 16 + l15, k);
  };
  auto mma = [&](int set, int i) {
    const int a = i / MI, b = (a & 1) ? MI - 1 - i % MI : i % MI;
    acc[a][b] = Mma<T>::mma(af[set][a], bf[set][b], acc[a][b]);
  };
  auto step = [&](int set, int nset, int nkk) {   // 4 MFMAs, the next step's fragments behind the first
    mma(set, 0); HBO_SB(); frags(nset, nkk); HBO_SB(); mma(set, 1); mma(set, 2); mma(set, 3);
  };
  frags(0, 0);

  const int* const yslot = job.yield_flag ? job.yield_flag + cu_token() : nullptr;
  int ypoll = 0;
  // slab s is in LDS; slab s + 1 waits in register set (s + 1) % PF and is replaced there by slab s + 1 + PF
  for (int s0 = 0; s0 + 1 < nk; s0 += PF) {
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const int s = s0 + u;
      if (s + 1 >= nk) break;
      const int set = (u + 1) % PF;
      if (yslot) {
        if (ypoll != 0)
          for (int spin = 0; spin < 256 && __hip_atomic_load(yslot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0; ++spin)
            __builtin_amdgcn_s_sleep(16);
        ypoll = __hip_atomic_load(yslot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
#pragma unroll
      for (int kk = 0; kk < KK - 2; ++kk) step(kk & 1, (kk + 1) & 1, kk + 1);
      mma(0, 0); HBO_SB(); frags(1, KK - 1); HBO_SB(); mma(0, 1);
      HBO_SB(); __syncthreads(); HBO_SB();          // every wave has read the slab's last fragments
      mma(0, 2); HBO_SB(); to_lds(set); HBO_SB(); mma(0, 3);
      if (s + 1 + PF < nk) fetch(set);
      mma(1, 0); mma(1, 1);
      HBO_SB(); __syncthreads(); HBO_SB();          // the next slab is in LDS
      mma(1, 2); HBO_SB(); frags(0, 0); HBO_SB(); mma(1, 3);
    }
  }
#pragma unroll
  for (int kk = 0; kk < KK - 1; ++kk) step(kk & 1, (kk + 1) & 1, kk + 1);
  mma(1, 0); mma(1, 1); mma(1, 2); mma(1, 3);
#undef HBO_SB

  if (job.C) {
#pragma unroll
    for (int a = 0; a < MI; ++a)
#pragma unroll
      for (int b = 0; b < MI; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = wm * WT + a * 16 + Mma<T>::crow(lane, r);
          const int col = wn * WT + b * 16 + l15;
          gst(job.C + (int64_t)row * job.ldc + col, job.alpha * acc[a][b][r]);
        }
  }
}
// which core a tile takes: the pipelined ones (128-tile: gemm_tile2, 64-tile: gemm_tile3; fp32 has 8 k steps per slab where fp64
// has 4, everything else is the same); gemm_tile stays for the A/B builds (-DHBO_GEMM_V1, -DHBO_GEMM_F32_V1, -DHBO_GEMM_NO64)
template <typename T, bool AKC, bool BKC, int TM>
__device__ __forceinline__ void run_tile(const TileJob<T>& job, unsigned char* smem) {
#ifndef HBO_GEMM_V1
#ifdef HBO_GEMM_F32_V1
  if constexpr (sizeof(T) == 4) gemm_tile<T, AKC, BKC, TM>(job, smem);
  else
#endif
  if constexpr (TM == 128) gemm_tile2<T, AKC, BKC, TM>(job, smem);
#ifndef HBO_GEMM_NO64
  else if constexpr (TM == 64) gemm_tile3<T, AKC, BKC, TM, HBO_PF64>(job, smem);
#endif
  else
#endif
    gemm_tile<T, AKC, BKC, TM>(job, smem);
}

// SYRK tiles by linear index (column-major over the trapezoid c in [c_lo,c_hi), r in [c,nrt)): used by
// the persistent form of the bulk trailing update, whose grid is smaller than the machine so that
// the panel kernels of the look-ahead (potf2 / trsm / next-column update) always find free CUs.
template <typename T, int TM>
__device__ __forceinline__ bool decode_syrk_linear(const GemmArgs& g, int tix, TileJob<T>& j) {
  constexpr int BKE = 128 / sizeof(T);
  constexpr int U = HBO_TILE / TM;
  const TaskDesc& t = g.tasks[blockIdx.z];
  const int nblk = t.nblk;
  const int nrt = (nblk + ((g.aug & 1) ? 1 : 0)) * U;
  const int chi = (g.c_hi < nblk ? g.c_hi : nblk) * U;
  int c = g.c_lo * U;
  while (c < chi && tix >= nrt - c) { tix -= nrt - c; ++c; }
  if (c >= chi) return false;
  const int r = c + tix;
  const int64_t ld = t.ld;
  T* Am = static_cast<T*>(t.A);
  j.A = Am + (int64_t)r * TM * ld + (int64_t)g.p0 * HBO_TILE;
  j.B = Am + (int64_t)c * TM * ld + (int64_t)g.p0 * HBO_TILE;
  j.C = Am + (int64_t)r * TM * ld + (int64_t)c * TM;
  j.lda = j.ldb = j.ldc = ld;
  j.colsq = nullptr;
  j.ksteps = g.kt * HBO_TILE / BKE;
  j.alpha = (T)-1; j.beta = 1;
  return true;
}

#ifdef HBO_GEMM_TIMING
__device__ unsigned long long hbo_dbg_gemm[4 * 8192];   // per workgroup of the traced launch: start, end, HW_ID, ksteps
int g_dbg_mode = -1, g_dbg_index = 0, g_dbg_seen = 0;   // host: trace the g_dbg_index-th launch of g_dbg_mode (+100: persistent)
#endif
template <typename T, bool AKC, bool BKC, int TM>
__device__ __forceinline__ void gemm_kernel_body(const GemmArgs& g, unsigned char* smem) {
  TileJob<T> job;
  job.yield_flag = g.yield_flag;
#ifdef HBO_GEMM_TIMING
  const int dbg_id = blockIdx.y * gridDim.x + blockIdx.x;
  const bool dbg = g.dbg && blockIdx.z == 0 && dbg_id < 8192 && threadIdx.x == 0;
  if (dbg) {
    hbo_dbg_gemm[4 * dbg_id] = wall_clock64();
    hbo_dbg_gemm[4 * dbg_id + 1] = 0;
    hbo_dbg_gemm[4 * dbg_id + 2] = ((unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) << 32) |
                                   (unsigned)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));
    hbo_dbg_gemm[4 * dbg_id + 3] = 0;
  }
#endif
  if (AKC && BKC && g.persistent) {
#ifdef HBO_GEMM_TIMING
    unsigned long long dbg_ks = 0;
#endif
    // Tiles are drawn from a shared counter when one is given: workgroup speeds differ by up to 1.8x (co-resident
    // panel-chain kernels, a lone workgroup on a CU runs faster), and with a static stride the slowest workgroup
    // with the most tiles set the kernel's duration (608 us against 460 us at the median speed).
    __shared__ int s_tix;
    for (int tix = blockIdx.x;; tix += gridDim.x) {
      if (g.work_counter) {
        if (threadIdx.x == 0) s_tix = atomicAdd(g.work_counter, 1);
        __syncthreads();
        tix = s_tix;
        __syncthreads();
      }
      if (TM == 128 && g.n_big > 0 && tix >= g.n_big) {
        // the tail of the launch on 64-tiles: big tile n_big + s / 4, quadrant s % 4
        const int s4 = tix - g.n_big;
        if (!decode_syrk_linear<T, 128>(g, g.n_big + (s4 >> 2), job)) break;
        const int qr = (s4 >> 1) & 1, qc = s4 & 1;
        const int64_t ldq = job.lda;
        job.A += (int64_t)qr * 64 * ldq;
        job.B += (int64_t)qc * 64 * ldq;
        job.C += (int64_t)qr * 64 * ldq + qc * 64;
        run_tile<T, AKC, BKC, 64>(job, smem);
        continue;
      }
      if (!decode_syrk_linear<T, TM>(g, tix, job)) break;
      run_tile<T, AKC, BKC, TM>(job, smem);
#ifdef HBO_GEMM_TIMING
      dbg_ks += (unsigned long long)job.ksteps;
#endif
    }
#ifdef HBO_GEMM_TIMING
    if (dbg) { hbo_dbg_gemm[4 * dbg_id + 1] = wall_clock64(); hbo_dbg_gemm[4 * dbg_id + 3] = dbg_ks; }
#endif
    return;
  }
  if (!(AKC && BKC) && g.persistent) {
    // Persistent form of the other modes (single matrix): g.pgx x g.pgy tiles in the order a one-tile-per-workgroup
    // launch would dispatch them (x fast, y -- which fixes the K length, longest first -- slow), drawn from a counter
    // by a grid that is smaller than the machine.  Used for the inverse products that co-run with the panel chain:
    // the whole grid is resident at once and leaves CUs free, so the chain's kernels never wait for a tile of up to
    // 256 K steps to finish before they get a slot (potf2 took 600 us once per evaluation, tools/trace_potrf.py).
    // Over a batch (ptasks > 1) the same counter runs over tiles x tasks, tile-major: the tiles of equal K of all tasks are
    // neighbours, and the launch stays slot-limited whatever the number of tasks.
    __shared__ int s_tix2;
    const int nt = g.ptasks > 1 ? g.ptasks : 1;
    const int total = g.pgx * g.pgy * nt;
    for (;;) {
      if (threadIdx.x == 0) s_tix2 = atomicAdd(g.work_counter, 1);
      __syncthreads();
      const int tix = s_tix2;
      __syncthreads();
      if (tix >= total) break;
      const int tile = tix / nt, task = g.ptasks > 1 ? tix % nt : (int)blockIdx.z;
      if (decode_job<T, TM>(g, job, tile % g.pgx, tile / g.pgx, g.pgx, task)) run_tile<T, AKC, BKC, TM>(job, smem);
    }
    return;
  }
  if (!decode_job<T, TM>(g, job, (int)blockIdx.x, (int)blockIdx.y, (int)gridDim.x, (int)blockIdx.z)) return;
  int ytok = 0;
  if (g.yield_mark && threadIdx.x == 0) ytok = yield_enter(g.yield_mark);
  run_tile<T, AKC, BKC, TM>(job, smem);
  if (g.yield_mark) {
    __syncthreads();
    if (threadIdx.x == 0) yield_leave(g.yield_mark, ytok);
  }
#ifdef HBO_GEMM_TIMING
  if (dbg) { hbo_dbg_gemm[4 * dbg_id + 1] = wall_clock64(); hbo_dbg_gemm[4 * dbg_id + 3] = (unsigned long long)job.ksteps; }
#endif
}

template <typename T, bool AKC, bool BKC, int TM>
__global__ __launch_bounds__(256, TM == 64 ? (sizeof(T) == 8 ? HBO_LB64 : 4) : 2) void gemm_kernel(GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  tl_begin(g.tl);
  gemm_kernel_body<T, AKC, BKC, TM>(g, smem);
  tl_end(g.tl);
}

#ifndef HBO_DEVICE_ONLY
template <typename T, bool AKC, bool BKC>
void launch_tile64(dim3 grid, int lds64, hipStream_t st, const GemmArgs& a) {
  // (round 6: the same tile on ONE wave -- a 64-thread workgroup with the 128-tile's 4 x 4 accumulator block, LDS-DMA staging, no barrier --
  //  was built, bit-identical, and slower everywhere: these launches are bound by the latency of a tile, and one SIMD then executes all
  //  64 MFMAs of a slab: N = 4096 2.57 -> 4.58 ms, N = 8192 10.65 -> 13.5, shard of 8 2.45 -> 3.55; profiles/r06_tile64_one_wave.md)
  hipLaunchKernelGGL((gemm_kernel<T, AKC, BKC, 64>), grid, dim3(256), lds64, st, a);
}
template <typename T>
void launch_gemm_t(const GemmArgs& a_in, dim3 grid, hipStream_t st) {
  GemmArgs a = a_in;
  // the persistent forms: SYRK (any tile size); TRTRI of a single matrix and the SWEEP modes (also over a batch) with a tile counter
  const bool sweep_mode = a.mode == GEMM_SWEEP_B || a.mode == GEMM_SWEEP_T || a.mode == GEMM_SWEEP_C;
  if (a.mode != GEMM_SYRK && !((a.mode == GEMM_TRTRI_A || a.mode == GEMM_TRTRI_B) && a.work_counter && grid.z == 1) && !(sweep_mode && a.work_counter) &&
      !(a.mode == GEMM_LAUUM && a.work_counter && grid.z == 1 && !a.small_tiles) && !(a.mode == GEMM_POST && a.work_counter && grid.z == 1))
    a.persistent = 0;
  // a resident grid needs at least one workgroup: on a small device / partition (or with a placement knob >= the CU count) the
  // callers' 2*(CUs - free) is <= 0 -- fall back to the plain grid AND drop the counter, or the kernel would take the counter branch
  // of a launch whose resident size is zero and compute no tile
  if (a.persistent <= 0) { a.persistent = 0; a.work_counter = nullptr; }
  a.ptasks = 0;
  // dynamic LDS of a 128-tile workgroup: one stage per operand for the pipelined cores, two for gemm_tile
#ifdef HBO_GEMM_V1
  int lds128 = GEMM_LDS_BYTES;
#else
#ifdef HBO_GEMM_F32_V1
  int lds128 = sizeof(T) == 8 ? 2 * OPERAND_BYTES : GEMM_LDS_BYTES;
#else
  int lds128 = 2 * OPERAND_BYTES;
#endif
#endif
  // 64-tiles keep the two-stage request although gemm_tile3 uses one: measured equal or better (N = 4096: 2.76 against 2.81 ms with
  // 20 KB, profiles/r04_gemm_pipeline.md); the persistent update runs its last round on 64-tiles inside the 128-tile kernel (n_big)
  const int lds64 = GEMM_LDS_BYTES_64;
  if (lds128 < lds64) lds128 = lds64;
#ifdef HBO_GEMM_DEBUG
  // HBO_GEMM_LDS=<bytes>: ask for more LDS than the kernel needs (above 80 KB: one workgroup per CU instead of two)
  static const int dbg_lds = getenv("HBO_GEMM_LDS") ? atoi(getenv("HBO_GEMM_LDS")) : 0;
  if (dbg_lds > lds128) lds128 = dbg_lds;
  const int attr128 = dbg_lds > GEMM_LDS_BYTES ? dbg_lds : GEMM_LDS_BYTES;
#else
  const int attr128 = GEMM_LDS_BYTES;
#endif
  static unsigned long long attr_seen = 0;
  if (hbo_first_use_on_device(attr_seen)) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_kernel<T, true, true, 128>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, attr128);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_kernel<T, true, false, 128>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, attr128);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_kernel<T, false, false, 128>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, attr128);
  }
  switch (a.mode) {
    case GEMM_SYRK:
      if (a.persistent > 0) {
        dim3 gp(a.persistent, 1, grid.z);
        if (a.small_tiles) launch_tile64<T, true, true>(gp, lds64, st, a);
        else hipLaunchKernelGGL((gemm_kernel<T, true, true, 128>), gp, dim3(256), lds128, st, a);
      } else if (a.small_tiles) {
        // 64x64 tiles: 4x the workgroups, a quarter of the per-tile latency -- for the skinny
        // updates on the critical path (next block column) and small trailing matrices
        dim3 g2(grid.x * 2, grid.y * 2, grid.z);
        launch_tile64<T, true, true>(g2, lds64, st, a);
      } else {
        hipLaunchKernelGGL((gemm_kernel<T, true, true, 128>), grid, dim3(256), lds128, st, a);
      }
      break;
    case GEMM_TRTRI_A:
    case GEMM_TRTRI_B:
      if (a.small_tiles && a.persistent > 0) {
        GemmArgs b = a; b.pgx = (int)grid.x * 2; b.pgy = (int)grid.y * 2;
        launch_tile64<T, true, false>(dim3(a.persistent, 1, 1), lds64, st, b);
      } else if (a.small_tiles) {
        dim3 g2(grid.x * 2, grid.y * 2, grid.z);
        launch_tile64<T, true, false>(g2, lds64, st, a);
      } else if (a.persistent > 0) {
        GemmArgs b = a; b.pgx = (int)grid.x; b.pgy = (int)grid.y;
        hipLaunchKernelGGL((gemm_kernel<T, true, false, 128>), dim3(a.persistent, 1, 1), dim3(256), lds128, st, b);
      } else {
        hipLaunchKernelGGL((gemm_kernel<T, true, false, 128>), grid, dim3(256), lds128, st, a);
      }
      break;
    case GEMM_SWEEP_B:
    case GEMM_SWEEP_T:
      // grid = tiles in 128-units (x, y) x tasks
      if (a.persistent > 0) {
        GemmArgs b = a; const int u = a.small_tiles ? 2 : 1;
        b.pgx = (int)grid.x * u; b.pgy = (int)grid.y * u; b.ptasks = (int)grid.z;
        if (a.small_tiles) launch_tile64<T, true, false>(dim3(a.persistent, 1, 1), lds64, st, b);
        else hipLaunchKernelGGL((gemm_kernel<T, true, false, 128>), dim3(a.persistent, 1, 1), dim3(256), lds128, st, b);
      } else if (a.small_tiles) {
        launch_tile64<T, true, false>(dim3(grid.x * 2, grid.y * 2, grid.z), lds64, st, a);
      } else {
        hipLaunchKernelGGL((gemm_kernel<T, true, false, 128>), grid, dim3(256), lds128, st, a);
      }
      break;
    case GEMM_SWEEP_C: {
      // grid.x = leading blocks b1 of the largest task: 1-D over their lower tiles
      const unsigned nt = a.small_tiles ? 2 * grid.x * (grid.x + 1) : grid.x * (grid.x + 1) / 2;
      if (a.persistent > 0) {
        GemmArgs b = a; b.pgx = (int)nt; b.pgy = 1; b.ptasks = (int)grid.z;
        if (a.small_tiles) launch_tile64<T, false, false>(dim3(a.persistent, 1, 1), lds64, st, b);
        else hipLaunchKernelGGL((gemm_kernel<T, false, false, 128>), dim3(a.persistent, 1, 1), dim3(256), lds128, st, b);
      } else if (a.small_tiles) {
        launch_tile64<T, false, false>(dim3(nt, 1, grid.z), lds64, st, a);
      } else {
        hipLaunchKernelGGL((gemm_kernel<T, false, false, 128>), dim3(nt, 1, grid.z), dim3(256), lds128, st, a);
      }
      break;
    }
    case GEMM_POST:
      if (a.persistent > 0) {   // (row tile, column tile) pairs drawn from a counter, long rows first -- see LAUUM
        GemmArgs b = a; b.pgx = (int)grid.x; b.pgy = (int)grid.y;
        hipLaunchKernelGGL((gemm_kernel<T, true, false, 128>), dim3(a.persistent, 1, 1), dim3(256), lds128, st, b);
      } else
      hipLaunchKernelGGL((gemm_kernel<T, true, false, 128>), grid, dim3(256), lds128, st, a);
      break;
    case GEMM_VTV:
#ifdef HBO_GEMM_DEBUG
      {
        // HBO_BENCH_VTV_KC=1 / 2: the same square launch through the <true,false> / <true,true> cores (operands read as k-contiguous
        // rows of the same buffer: meaningless numbers, the loop's rate for those LDS layouts)
        static const int kc = getenv("HBO_BENCH_VTV_KC") ? atoi(getenv("HBO_BENCH_VTV_KC")) : 0;
        if (kc == 1) { hipLaunchKernelGGL((gemm_kernel<T, true, false, 128>), grid, dim3(256), lds128, st, a); break; }
        if (kc == 2) { hipLaunchKernelGGL((gemm_kernel<T, true, true, 128>), grid, dim3(256), lds128, st, a); break; }
      }
#endif
      hipLaunchKernelGGL((gemm_kernel<T, false, false, 128>), grid, dim3(256), lds128, st, a);
      break;
    case GEMM_LAUUM:
      if (a.small_tiles) {
        dim3 g2(2 * grid.x * (grid.x + 1), 1, grid.z);
        launch_tile64<T, false, false>(g2, lds64, st, a);
      } else {
        // 1-D grid over the lower tiles (grid.x = block count of the largest task)
        dim3 g1(grid.x * (grid.x + 1) / 2, 1, grid.z);
        if (a.persistent > 0) {
          // the same tiles in the same order, drawn from a counter by a resident grid: the hardware deals the workgroups of a
          // plain grid to the 8 XCDs in turn and waits when the next one's XCD is full -- with tiles of 1 to 64 K blocks a tenth
          // of the slots stood empty in the middle of the launch (in-kernel stamps: 459 of 512 busy on average)
          GemmArgs b = a; b.pgx = (int)g1.x; b.pgy = 1;
          hipLaunchKernelGGL((gemm_kernel<T, false, false, 128>), dim3(a.persistent, 1, 1), dim3(256), lds128, st, b);
        } else
        hipLaunchKernelGGL((gemm_kernel<T, false, false, 128>), g1, dim3(256), lds128, st, a);
      }
      break;
  }
}

#endif  // HBO_DEVICE_ONLY
}  // namespace

#ifndef HBO_DEVICE_ONLY
#ifdef HBO_GEMM_TIMING
extern "C" void hbo_dbg_gemm_wall(unsigned long long* host, int mode, int index) {
  if (host) { hipMemcpyFromSymbol(host, HIP_SYMBOL(hbo_dbg_gemm), sizeof(unsigned long long) * 4 * 8192); return; }
  static unsigned long long zeros[4 * 8192];
  hipMemcpyToSymbol(HIP_SYMBOL(hbo_dbg_gemm), zeros, sizeof zeros);
  g_dbg_mode = mode; g_dbg_index = index; g_dbg_seen = 0;
}
#endif
void launch_gemm(int dtype, const GemmArgs& a_in, dim3 grid, hipStream_t st) {
  GemmArgs a = a_in;
#ifdef HBO_GEMM_TIMING
  a.dbg = 0;
  if ((a.persistent ? a.mode + 100 : a.mode) == g_dbg_mode) a.dbg = (g_dbg_seen++ == g_dbg_index);
#endif
  if (dtype == HBO_F64) launch_gemm_t<double>(a, grid, st);
  else launch_gemm_t<float>(a, grid, st);
}
#endif  // HBO_DEVICE_ONLY
