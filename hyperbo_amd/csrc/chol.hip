// Diagonal-panel kernels of the blocked right-looking Cholesky on gfx950:
//   potf2_kernel  : LDS-resident 128x128 factorisation of the diagonal block (one workgroup),
//                   plus the inverses of its eight 16x16 leaf blocks (written to W).
//   trsm_kernel   : X = A_panel * L_pp^-T by blocked forward substitution on MFMA, one wave per
//                   16 panel rows, L_pp and the leaf inverses staged in LDS as packed 16x16 tiles.
//                   With IDENT=true the same code produces W_pp = L_pp^-1 (trtri base case).
// Replaces what jax.scipy.linalg.cholesky lowers to (hyperbo/basics/linalg.py:31,134).
#include "hbo_internal.h"
#include <type_traits>
#include <limits.h>
#include <string.h>

namespace {

#include "panel_dev.h"
// LDS-resident factorisation of one 128x128 diagonal block as 36 packed 16x16 lower tiles (78 KB
// for fp64, so the workgroup fits on a CU next to a running GEMM workgroup):
//   leaf   : wave 0 factors the symmetric diagonal tile on MFMA (leaf_cholesky),
//   (B)    : rows below the leaf: X = A M^T on MFMA with M = leaf^-1 (built inside leaf_cholesky),
//   (C)    : rank-16 MFMA update of the trailing tiles; wave 0 takes the next diagonal tile first and
//            factors it while waves 1-3 finish the rest (the leaf chain is the critical path).
// The leaf inverses also go to W (used by trsm_kernel and the trtri base case).
#ifdef HBO_POTF2_TIMING
__device__ unsigned long long hbo_dbg_stamps[64];
__device__ unsigned long long hbo_dbg_wall[3 * 256];   // per panel: start, end (100 MHz s_memrealtime), HW_ID
__device__ unsigned long long hbo_dbg_trsm[3 * 128];   // trsm of panel hbo_dbg_trsm_panel: per workgroup start, end, HW_ID
__device__ int hbo_dbg_trsm_panel = 36;
#define STAMP(i) do { if (threadIdx.x == 0) hbo_dbg_stamps[i] = __builtin_readcyclecounter(); } while (0)
#else
#define STAMP(i) do {} while (0)
#endif
// workgroup of potf2: eight waves (two per SIMD) -- wave 0 owns the leaf chain, seven share the solves and the trailing tiles
// (with four, the first two steps of a full block ended 3600 / 2100 cycles after wave 0's leaf; chol_bench)
#ifndef HBO_POTF2_WAVES
#define HBO_POTF2_WAVES 8
#endif
constexpr int POTF2_WAVES = HBO_POTF2_WAVES;
constexpr int POTF2_THREADS = 64 * POTF2_WAVES;
template <typename T>
__device__ __forceinline__ void potf2_body(const TaskDesc& t, int p, int* info_slot, unsigned char* smem) {
  typedef typename Mma<T>::acc_t acc_t;
  T* sT = reinterpret_cast<T*>(smem);       // 36 tiles [16][17]
  T* sM = sT + 36 * TILE_ELEMS;             // inverse of the current leaf [16][17]
  T* sDinv = sM + TILE_ELEMS;               // [128] 1 / diag(L)
  const int64_t ld = t.ld;
  T* Ab = static_cast<T*>(t.A) + (int64_t)p * NB * ld + (int64_t)p * NB;
  T* Wb = static_cast<T*>(t.W) + (int64_t)p * NB * ld + (int64_t)p * NB;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, lq = lane >> 4;
  const int ei = (tid & 255) >> 4, ej = tid & 15;    // element of a tile owned by this thread for I/O
  STAMP(0);
#ifdef HBO_POTF2_TIMING
  if (tid == 0 && blockIdx.x == 0 && p < 256) {
    hbo_dbg_wall[3 * p] = wall_clock64();
    hbo_dbg_wall[3 * p + 2] = ((unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) << 32) |   // XCC_ID
                              (unsigned)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));                        // HW_ID
  }
#endif

  auto factor_leaf = [&](int jb, const acc_t* from_regs = nullptr) {   // wave 0 only
    T* dt = sT + tri_index(jb, jb) * TILE_ELEMS;
    acc_t acc;
    if (from_regs) acc = *from_regs;
    else {
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[r] = dt[Mma<T>::crow(lane, r) * TS + l15];
    }
    const int bad = leaf_cholesky4<T>(acc, dt, sM, sDinv + jb * 16, Wb + (int64_t)(jb * 16) * ld + jb * 16, ld, lane);
    if (bad >= 0 && lane == 0) atomicMin(info_slot, p * NB + jb * 16 + bad + 1);
  };
  auto solve_tile = [&](int jb, int R) {   // rows of tile (R, jb): X = A M^T, in place
    T* xt = sT + tri_index(R, jb) * TILE_ELEMS;
    acc_t acc = {0, 0, 0, 0};
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) acc = Mma<T>::mma(xt[l15 * TS + kk * 4 + lq], sM[l15 * TS + kk * 4 + lq], acc);
#pragma unroll
    for (int r = 0; r < 4; ++r) xt[Mma<T>::crow(lane, r) * TS + l15] = acc[r];
  };
  auto update_tile = [&](int jb, int I, int J) {   // C[I][J] -= X_I X_J^T (K = 16)
    T* ct = sT + tri_index(I, J) * TILE_ELEMS;
    const T* at = sT + tri_index(I, jb) * TILE_ELEMS;
    const T* bt = sT + tri_index(J, jb) * TILE_ELEMS;
    acc_t acc;
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = ct[Mma<T>::crow(lane, r) * TS + l15];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) acc = Mma<T>::mma(-at[l15 * TS + kk * 4 + lq], bt[l15 * TS + kk * 4 + lq], acc);
#pragma unroll
    for (int r = 0; r < 4; ++r) ct[Mma<T>::crow(lane, r) * TS + l15] = acc[r];
  };

  auto update_tile2 = [&](int jb, int I0, int J0, int I1, int J1) {   // two independent tiles, interleaved
    T* c0 = sT + tri_index(I0, J0) * TILE_ELEMS;
    T* c1 = sT + tri_index(I1, J1) * TILE_ELEMS;
    const T* a0 = sT + tri_index(I0, jb) * TILE_ELEMS;
    const T* b0 = sT + tri_index(J0, jb) * TILE_ELEMS;
    const T* a1 = sT + tri_index(I1, jb) * TILE_ELEMS;
    const T* b1 = sT + tri_index(J1, jb) * TILE_ELEMS;
    acc_t x0, x1;
    T fa0[4], fb0[4], fa1[4], fb1[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { x0[r] = c0[Mma<T>::crow(lane, r) * TS + l15]; x1[r] = c1[Mma<T>::crow(lane, r) * TS + l15]; }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      fa0[kk] = -a0[l15 * TS + kk * 4 + lq]; fb0[kk] = b0[l15 * TS + kk * 4 + lq];
      fa1[kk] = -a1[l15 * TS + kk * 4 + lq]; fb1[kk] = b1[l15 * TS + kk * 4 + lq];
    }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) { x0 = Mma<T>::mma(fa0[kk], fb0[kk], x0); x1 = Mma<T>::mma(fa1[kk], fb1[kk], x1); }
#pragma unroll
    for (int r = 0; r < 4; ++r) { c0[Mma<T>::crow(lane, r) * TS + l15] = x0[r]; c1[Mma<T>::crow(lane, r) * TS + l15] = x1[r]; }
  };

  // leaves that hold data: the identity padding of the last block factors to itself (a matrix of 64 points has four
  // of its eight leaves empty); their leaf inverse is the identity, written here once
  int nleaf = 8;
  {
    const int64_t rows = (int64_t)t.n - (int64_t)p * NB;
    if (rows < NB) nleaf = rows <= 0 ? 0 : (int)((rows + 15) / 16);
    for (int jb = nleaf + wave; jb < 8; jb += POTF2_WAVES)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = Mma<T>::crow(lane, r);
        gst(Wb + (int64_t)(jb * 16 + row) * ld + jb * 16 + l15, row == l15 ? (T)1 : (T)0);
      }
    if (tid < NB && tid >= nleaf * 16) sDinv[tid] = (T)1;
  }
  // The first diagonal tile goes straight from memory into wave 0's MFMA layout (both triangles from the lower one) and is
  // factored while the other 35 tiles are still on their way to LDS: the first leaf (4500 cycles) hides behind the load phase
  // (7300) instead of following it.  Tile 0's place in LDS is written by the leaf only (its columns of L).
  const bool lead_tile0 = nleaf > 0;
  acc_t tile0 = {0, 0, 0, 0};
  if (wave == 0 && lead_tile0) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = Mma<T>::crow(lane, r);
      const int rr = row > l15 ? row : l15, cc = row > l15 ? l15 : row;
      tile0[r] = gld(Ab + (int64_t)rr * ld + cc);
    }
  }
  // load: one element of every lower tile of this thread's share of the 36 (256-thread group `sh` takes the tiles == sh mod NSH;
  // the share is a compile-time parameter of the loop body: a run-time test per tile serialised the 36 loads, 21 000 cycles
  // instead of 6 000); diagonal tiles are mirrored to full symmetry
  constexpr int NSH = POTF2_THREADS / 256;
  const int sh = __builtin_amdgcn_readfirstlane(tid >> 8);
  T vload[36 / NSH];
  auto load_issue = [&](auto share) {
    constexpr int SH = decltype(share)::value;
    int tile = 0;
#pragma unroll
    for (int I = 0; I < 8; ++I)
#pragma unroll
      for (int J = 0; J <= I; ++J, ++tile) {
        if (tile % NSH != SH) continue;
        if (tile == 0 && lead_tile0) { vload[0] = (T)0; continue; }   // (wave 0 fetched it in the MFMA layout, above)
        int r = ei, c = ej;
        if (I == J && ej > ei) { r = ej; c = ei; }
        vload[tile / NSH] = gld(Ab + (int64_t)(I * 16 + r) * ld + J * 16 + c);
      }
  };
  auto load_commit = [&](auto share) {
    constexpr int SH = decltype(share)::value;
#pragma unroll
    for (int k = 0; k < 36 / NSH; ++k)
      if (!(k == 0 && SH == 0 && lead_tile0)) sT[(k * NSH + SH) * TILE_ELEMS + ei * TS + ej] = vload[k];
  };
  static_assert(NSH == 1 || NSH == 2, "one or two 256-thread groups");
  if (NSH == 1 || sh == 0) load_issue(std::integral_constant<int, 0>());
  else load_issue(std::integral_constant<int, NSH - 1>());
  // wave 0: the leaf first (its tile arrives first), then its part of the other tiles to LDS -- the others wait for the leaf anyway
  if (wave == 0 && lead_tile0) factor_leaf(0, &tile0);
  if (NSH == 1 || sh == 0) load_commit(std::integral_constant<int, 0>());
  else load_commit(std::integral_constant<int, NSH - 1>());
  STAMP(1);
  __syncthreads();
  STAMP(2);
  for (int jb = 0; jb < nleaf; ++jb) {
    // ---- (B) rows below the leaf: X = A M^T, one 16-row tile per wave and pass (wave 0 takes tile jb+1, whose
    //      result it needs first in (C)) ------------------------------------------------------
    for (int R = jb + 1 + wave; R < nleaf; R += POTF2_WAVES) solve_tile(jb, R);   // (tiles of padding rows are zero)
    __syncthreads();
    STAMP(3 + 3 * jb);
    if (jb == nleaf - 1) break;   // (the rows below the last data leaf are zero: (B) left them zero)
    // ---- (C) trailing update; wave 0 owns the next diagonal tile and factors it right away ----
    {
      const int m = nleaf - 1 - jb;
      const int ntiles = m * (m + 1) / 2;            // tile 0 is (jb+1, jb+1): wave 0's
      // the leaf costs about four tile updates: while the other waves have more than that each (four waves: more than ~13 other
      // tiles, the first two steps of a full block) wave 0 takes the last n0 of them after its leaf, so that all waves finish together
      // helper waves: all but wave 0 -- and, with eight waves, but wave 4, which shares wave 0's SIMD (the leaf chain beside a
      // second wave's LDS round trips and MFMAs: 4900-5000 cycles instead of 4400)
      constexpr int NH = POTF2_WAVES == 8 ? 6 : POTF2_WAVES - 1;
      const int hw = POTF2_WAVES == 8 ? (wave < 4 ? wave : wave - 1) : wave;   // 1..NH for the helpers
      const int n0 = (ntiles - 1 > 4 * NH + 1) ? (ntiles - 1 - (4 * NH + 1)) / POTF2_WAVES : 0;
      const int nshared = ntiles - n0;               // tiles [1, nshared) go round the waves 1..3
      auto tile_of = [&](int tix, int& I, int& J) {
        int ii = 0;
        while ((ii + 1) * (ii + 2) / 2 <= tix) ++ii;
        I = jb + 1 + ii; J = jb + 1 + tix - ii * (ii + 1) / 2;
      };
      if (wave == 0) {
        update_tile(jb, jb + 1, jb + 1);
        factor_leaf(jb + 1);
        STAMP(4 + 3 * jb);
        for (int tix = nshared; tix < ntiles; ++tix) { int I, J; tile_of(tix, I, J); update_tile(jb, I, J); }
      } else if (!(POTF2_WAVES == 8 && wave == 4)) {
        // two tiles per pass: their LDS round trips and MFMAs interleave (one tile at a time ran at the latency of its
        // own load -> 4 MFMA -> store chain, ~1000 cycles per tile)
        int tix = hw;
        for (; tix + NH < nshared; tix += 2 * NH) {
          int I0, J0, I1, J1;
          tile_of(tix, I0, J0); tile_of(tix + NH, I1, J1);
          update_tile2(jb, I0, J0, I1, J1);
        }
        if (tix < nshared) { int I, J; tile_of(tix, I, J); update_tile(jb, I, J); }
      }
    }
    __syncthreads();
    STAMP(5 + 3 * jb);
  }
  STAMP(30);

  // write L (lower; zeros above the diagonal inside the diagonal tiles)
  auto store_share = [&](auto share) {
    constexpr int SH = decltype(share)::value;
    int tile = 0;
#pragma unroll
    for (int I = 0; I < 8; ++I)
#pragma unroll
      for (int J = 0; J <= I; ++J, ++tile) {
        if (tile % NSH != SH) continue;
        T v = sT[tile * TILE_ELEMS + ei * TS + ej];
        if (I == J && ej > ei) v = (T)0;
        gst(Ab + (int64_t)(I * 16 + ei) * ld + J * 16 + ej, v);
      }
  };
  if (NSH == 1 || sh == 0) store_share(std::integral_constant<int, 0>());
  else store_share(std::integral_constant<int, NSH - 1>());
  STAMP(31);
  STAMP(32);
#ifdef HBO_POTF2_TIMING
  if (tid == 0 && blockIdx.x == 0 && p < 256) hbo_dbg_wall[3 * p + 1] = wall_clock64();
#endif
}
template <typename T>
__global__ __launch_bounds__(POTF2_THREADS) void potf2_kernel(const TaskDesc* tasks, int p, int* info, int* yield_flag, unsigned long long* tl) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __builtin_amdgcn_s_setprio(3);            // critical path: outrank co-resident GEMM waves
  const TaskDesc& t = tasks[blockIdx.x];
  if (p >= t.nblk) return;
  tl_begin(tl);
  // single matrix: count this workgroup into the yield table entry of its CU -- background GEMM workgroups that share the
  // CU pause at their next K step (potf2's small MFMAs queue behind their 64-cycle ones and its LDS traffic behind theirs:
  // 50 us beside them, 22 us alone)
  int tok = 0;
  if (yield_flag && threadIdx.x == 0) tok = yield_enter(yield_flag);
  potf2_body<T>(t, p, info + blockIdx.x, smem);
  if (yield_flag) {
    __syncthreads();
    if (threadIdx.x == 0) yield_leave(yield_flag, tok);
  }
  tl_end(tl);
}

template <typename T>
constexpr int trsm_tile() { return sizeof(T) == 8 ? 256 : TILE_ELEMS; }
// element (r, c) of a 16 x 16 tile of the panel solve.  fp64: unpadded rows, the column XOR-ed with 2 (r / 2) -- every access
// pattern of trsm_body (one row per 16 lanes; MFMA operand fragments: 16 rows x 2 columns per half-wave; accumulator layout:
// 2 rows x 16 columns per half-wave) then covers all banks once, and the 40 tiles take exactly 80 KB: TWO workgroups fit on a
// CU the bulk update leaves free (with padded 17-column rows, 87 KB, one did: 35 us per panel solve beside the bulk update,
// 10-13 us alone).  fp32 keeps the padded rows (43.5 KB).
template <typename T>
__device__ __forceinline__ int trsm_at(int r, int c) { return sizeof(T) == 8 ? r * 16 + (c ^ ((r >> 1) << 1)) : r * TS + c; }
template <typename T>
constexpr int trsm_lds_bytes() { return (28 + 8 + 4) * trsm_tile<T>() * (int)sizeof(T); }   // fp64: 80 KB (also fits beside one GEMM workgroup)
__device__ __forceinline__ int strict_index(int I, int J) { return I * (I - 1) / 2 + J; }

// grid.x = 64-row group, grid.y = (IDENT ? diagonal block p : unused), grid.z = task
// 64 rows [row0, row0 + 64) of the panel (or of the identity for IDENT) against diagonal block p.  `stage`: load
// L_pp and its leaf inverses into LDS first -- a caller that walks several row groups of the same panel (chain.hip)
// stages once.
template <typename T, bool IDENT>
__device__ __forceinline__ void trsm_body(const TaskDesc& t, int p, int64_t row0, unsigned char* smem, bool stage,
                                          unsigned short* xp = nullptr, int nkb = 0, int kb_off = 0) {
  typedef typename Mma<T>::acc_t acc_t;
  constexpr int TE = trsm_tile<T>();
  T* sLt = reinterpret_cast<T*>(smem);          // 28 packed strictly-lower tiles of L_pp
  T* sWi = sLt + 28 * TE;                       // 8 leaf inverses (stand in for the diagonal tiles)
  T* sSc = sWi + 8 * TE;                        // 4 per-wave scratch tiles
  const int64_t ld = t.ld;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, lq = lane >> 4;
  const T* Lb = static_cast<const T*>(t.A) + (int64_t)p * NB * ld + (int64_t)p * NB;
  const T* Wb = static_cast<const T*>(t.W) + (int64_t)p * NB * ld + (int64_t)p * NB;

  const int64_t wrow0 = row0 + wave * 16;                 // this wave's 16 rows
  const int rg = IDENT ? (int)((wrow0 - (int64_t)p * NB) >> 4) : 0;  // row group inside the block
  T* Ap = IDENT ? nullptr : static_cast<T*>(t.A) + wrow0 * ld + (int64_t)p * NB;
  T* Wout = IDENT ? static_cast<T*>(t.W) + (int64_t)p * NB * ld + (int64_t)p * NB : nullptr;

  // the wave's 16 x 128 panel rows, all eight tiles in flight before L_pp is staged
  acc_t areg[8];
#pragma unroll
  for (int jb = 0; jb < 8; ++jb)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = Mma<T>::crow(lane, r);
      if (IDENT) areg[jb][r] = (rg == jb && row == l15) ? (T)1 : (T)0;
      else areg[jb][r] = gld(Ap + (int64_t)row * ld + jb * 16 + l15);
    }
  // stage L_pp (lower tiles) and the leaf inverses: thread = one element of each 16x16 tile
  if (stage) {
    const int i = tid >> 4, j = tid & 15;
    int tile = 0;
#pragma unroll
    for (int I = 1; I < 8; ++I)
#pragma unroll
      for (int J = 0; J < I; ++J, ++tile)
        sLt[tile * TE + trsm_at<T>(i, j)] = gld(Lb + (int64_t)(I * 16 + i) * ld + J * 16 + j);
#pragma unroll
    for (int b = 0; b < 8; ++b)
      sWi[b * TE + trsm_at<T>(i, j)] = gld(Wb + (int64_t)(b * 16 + i) * ld + b * 16 + j);
  }
  if (stage) __syncthreads();

  T* sc = sSc + wave * TE;   // wave-private: LDS ops of one wave execute in order
  T xneg[8][4];                      // -X in A-operand layout, per 16-column block
#pragma unroll
  for (int jb = 0; jb < 8; ++jb) {
    // acc = A[:,jb] - sum_{kb<jb} X[:,kb] * L[jb,kb]^T   (two accumulators: shorter MFMA chain)
    acc_t acc = areg[jb];
    acc_t acc2 = (acc_t){0, 0, 0, 0};
#pragma unroll
    for (int kb = 0; kb < 8; ++kb) {
      if (kb < jb) {
        const T* lt = sLt + strict_index(jb, kb) * TE;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          const T bfrag = lt[trsm_at<T>(l15, kk * 4 + lq)];   // B[k][j] = L[jb*16+j][kb*16+k]
          if (kb & 1) acc2 = Mma<T>::mma(xneg[kb][kk], bfrag, acc2);
          else acc = Mma<T>::mma(xneg[kb][kk], bfrag, acc);
        }
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] += acc2[r];
    // T (C layout) -> scratch -> A layout
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int r = 0; r < 4; ++r) sc[trsm_at<T>(Mma<T>::crow(lane, r), l15)] = acc[r];
    __builtin_amdgcn_wave_barrier();
    T ta[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) ta[kk] = sc[trsm_at<T>(l15, kk * 4 + lq)];
    // X[:,jb] = T * Winv_jb^T
    acc_t x = (acc_t){0, 0, 0, 0};
    const T* wt = sWi + jb * TE;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const T bfrag = wt[trsm_at<T>(l15, kk * 4 + lq)];       // B[k][j] = Winv[j][k]
      x = Mma<T>::mma(ta[kk], bfrag, x);
    }
    // store
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = Mma<T>::crow(lane, r);
      if (IDENT) {
        // W = X^T ; the diagonal leaves were already written by potf2
        if (rg != jb) gst(Wout + (int64_t)(jb * 16 + l15) * ld + rg * 16 + row, x[r]);
      } else {
        gst(Ap + (int64_t)row * ld + jb * 16 + l15, x[r]);
        if constexpr (sizeof(T) == 4) {
          if (xp) {   // the same value as three bf16 planes (see SplitOut)
            const int64_t grow = wrow0 + row;
            unsigned short h, m, l;
            hbo_split3((float)x[r], h, m, l);
            unsigned short* o = xp + ((grow / NB * nkb + kb_off + jb) * 3) * (int64_t)(NB * 16) + (grow % NB) * 16 + l15;
            o[0] = h; o[NB * 16] = m; o[2 * NB * 16] = l;
          }
        }
      }
    }
    if (jb < 7) {
      // X (C layout) -> scratch -> negated A layout for later column blocks
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int r = 0; r < 4; ++r) sc[trsm_at<T>(Mma<T>::crow(lane, r), l15)] = x[r];
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) xneg[jb][kk] = -sc[trsm_at<T>(l15, kk * 4 + lq)];
    }
  }
}
template <typename T, bool IDENT>
__global__ __launch_bounds__(256) void trsm_kernel(const TaskDesc* tasks, int p_arg, int* yield_tab, SplitOut so, unsigned long long* tl) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __builtin_amdgcn_s_setprio(3);                // critical path: outrank co-resident GEMM waves
  const TaskDesc& t = tasks[blockIdx.z];
  const int p = IDENT ? p_arg + (int)blockIdx.y : p_arg;   // IDENT: p_arg = first diagonal block
  if (p >= t.nblk) return;
#ifdef HBO_POTF2_TIMING
  const bool dbg = !IDENT && p == hbo_dbg_trsm_panel && blockIdx.x < 128 && threadIdx.x == 0;
  if (dbg) {
    hbo_dbg_trsm[3 * blockIdx.x] = wall_clock64();
    hbo_dbg_trsm[3 * blockIdx.x + 2] = ((unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) << 32) |
                                       (unsigned)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));
  }
#endif
  // first row handled by this workgroup
  int64_t row0;
  if (IDENT) {
    row0 = (int64_t)p * NB + (int64_t)blockIdx.x * 64;            // blockIdx.x in {0,1}
  } else {
    const int64_t first = (int64_t)(p + 1) * NB;
    const int64_t nrows = (int64_t)(t.nblk + 1) * NB - first;     // incl. augmented tile-row
    // workgroup i runs on XCD (i + const) mod 8 whatever the load: in a ragged batch the rows that exist are the low ones in
    // every task, so the row index is rotated by the task index to spread them over the XCDs
    const int64_t bx = ((int64_t)blockIdx.x + 3 * (int64_t)blockIdx.z) % gridDim.x;
    if (bx * 64 >= nrows) return;
    row0 = first + bx * 64;
  }
  int tok = 0;
  tl_begin(tl);
  if (yield_tab && threadIdx.x == 0) tok = yield_enter(yield_tab);
  trsm_body<T, IDENT>(t, p, row0, smem, true, so.xp ? so.xp + (int64_t)blockIdx.z * so.task_stride : nullptr, so.nkb, so.kb_off);
  if (yield_tab) {
    __syncthreads();
    if (threadIdx.x == 0) yield_leave(yield_tab, tok);
  }
  tl_end(tl);
#ifdef HBO_POTF2_TIMING
  if (dbg) hbo_dbg_trsm[3 * blockIdx.x + 1] = wall_clock64();
#endif
}

#ifndef HBO_DEVICE_ONLY
template <typename T>
void set_attrs() {
  static unsigned long long seen = 0;
  if (!hbo_first_use_on_device(seen)) return;
  hipFuncSetAttribute(reinterpret_cast<const void*>(&potf2_kernel<T>),
                      hipFuncAttributeMaxDynamicSharedMemorySize, potf2_lds_bytes<T>());
  hipFuncSetAttribute(reinterpret_cast<const void*>(&trsm_kernel<T, false>),
                      hipFuncAttributeMaxDynamicSharedMemorySize, trsm_lds_bytes<T>());
  hipFuncSetAttribute(reinterpret_cast<const void*>(&trsm_kernel<T, true>),
                      hipFuncAttributeMaxDynamicSharedMemorySize, trsm_lds_bytes<T>());
}

template <typename T>
void potf2_t(const TaskDesc* tasks, int ntasks, int p, int* info, hipStream_t st, int* yield_flag, unsigned long long* tl) {
  set_attrs<T>();
  hipLaunchKernelGGL((potf2_kernel<T>), dim3(ntasks), dim3(POTF2_THREADS), potf2_lds_bytes<T>(), st, tasks, p, info,
                     yield_flag, tl);
}
template <typename T>
void trsm_t(const TaskDesc* tasks, int ntasks, int p, int max_nblk, hipStream_t st, int* yield_tab, const SplitOut& so, unsigned long long* tl) {
  set_attrs<T>();
  const int nrows = (max_nblk + 1 - (p + 1)) * NB;
  if (nrows <= 0) return;
  hipLaunchKernelGGL((trsm_kernel<T, false>), dim3(nrows / 64, 1, ntasks), dim3(256), trsm_lds_bytes<T>(), st,
                     tasks, p, yield_tab, so, tl);
}
template <typename T>
void trtri_diag_t(const TaskDesc* tasks, int ntasks, int p_lo, int p_hi, hipStream_t st) {
  set_attrs<T>();
  if (p_hi <= p_lo) return;
  SplitOut none; memset(&none, 0, sizeof none);
  hipLaunchKernelGGL((trsm_kernel<T, true>), dim3(2, p_hi - p_lo, ntasks), dim3(256), trsm_lds_bytes<T>(), st,
                     tasks, p_lo, (int*)nullptr, none, (unsigned long long*)nullptr);
}

#endif  // HBO_DEVICE_ONLY
}  // namespace

#ifndef HBO_DEVICE_ONLY
#ifdef HBO_POTF2_TIMING
void dbg_read_stamps(unsigned long long* host) { hipMemcpyFromSymbol(host, HIP_SYMBOL(hbo_dbg_stamps), sizeof(unsigned long long) * 64); }
extern "C" void hbo_dbg_trsm_wall(unsigned long long* host, int panel) {
  if (host) hipMemcpyFromSymbol(host, HIP_SYMBOL(hbo_dbg_trsm), sizeof(unsigned long long) * 3 * 128);
  else hipMemcpyToSymbol(HIP_SYMBOL(hbo_dbg_trsm_panel), &panel, sizeof(int));
}
extern "C" void hbo_dbg_potf2_wall(unsigned long long* host) { hipMemcpyFromSymbol(host, HIP_SYMBOL(hbo_dbg_wall), sizeof(unsigned long long) * 3 * 256); }
#endif
void launch_potf2(int dtype, const TaskDesc* tasks, int ntasks, int p, int* info, hipStream_t st, int* yield_flag, unsigned long long* tl) {
  if (dtype == HBO_F64) potf2_t<double>(tasks, ntasks, p, info, st, yield_flag, tl);
  else potf2_t<float>(tasks, ntasks, p, info, st, yield_flag, tl);
}
void launch_trsm(int dtype, const TaskDesc* tasks, int ntasks, int p, int max_nblk, hipStream_t st, int* yield_tab, const SplitOut* so,
                 unsigned long long* tl) {
  SplitOut o; memset(&o, 0, sizeof o);
  if (so) o = *so;
  if (dtype == HBO_F64) trsm_t<double>(tasks, ntasks, p, max_nblk, st, yield_tab, o, tl);
  else trsm_t<float>(tasks, ntasks, p, max_nblk, st, yield_tab, o, tl);
}
void launch_trtri_diag(int dtype, const TaskDesc* tasks, int ntasks, int p_lo, int p_hi, hipStream_t st) {
  if (dtype == HBO_F64) trtri_diag_t<double>(tasks, ntasks, p_lo, p_hi, st);
  else trtri_diag_t<float>(tasks, ntasks, p_lo, p_hi, st);
}
#endif  // HBO_DEVICE_ONLY
