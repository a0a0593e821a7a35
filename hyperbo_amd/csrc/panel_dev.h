// Device helpers of the diagonal-panel kernels (chol.hip: potf2_kernel / trsm_kernel) shared with the single-workgroup evaluation
// of small tasks (small.hip): the MFMA tile traits, cross-lane moves, the packed 16x16 LDS tile layout and the four-columns-per-step
// leaf factorisation that also yields the leaf's inverse.  Include INSIDE an anonymous namespace, after hbo_internal.h,
// <type_traits> and <limits.h>.
#pragma once

template <typename T> struct Mma;
template <> struct Mma<double> {
  typedef double acc_t __attribute__((ext_vector_type(4)));
  static __device__ __forceinline__ acc_t mma(double a, double b, acc_t c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ int crow(int lane, int r) { return (lane >> 4) + 4 * r; }
};
template <> struct Mma<float> {
  typedef float acc_t __attribute__((ext_vector_type(4)));
  static __device__ __forceinline__ acc_t mma(float a, float b, acc_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ int crow(int lane, int r) { return 4 * (lane >> 4) + r; }
};

constexpr int NB = HBO_TILE;   // 128

__device__ __forceinline__ double readlane_t(double v, int lane) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_readlane(lo, lane);
  hi = __builtin_amdgcn_readlane(hi, lane);
  return __hiloint2double(hi, lo);
}
// value of the lane 16 away inside each 32-lane half (ds_swizzle bit mode: and 0x1f, or 0, xor 0x10)
__device__ __forceinline__ double swap16(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_ds_swizzle(lo, 0x401F);
  hi = __builtin_amdgcn_ds_swizzle(hi, 0x401F);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ float swap16(float v) {
  return __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), 0x401F));
}
__device__ __forceinline__ float readlane_t(float v, int lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
// value held by lane `src` (any lane -> any lane, through the LDS crossbar)
__device__ __forceinline__ double lane_gather(double v, int src) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_ds_bpermute(src << 2, lo);
  hi = __builtin_amdgcn_ds_bpermute(src << 2, hi);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ float lane_gather(float v, int src) {
  return __int_as_float(__builtin_amdgcn_ds_bpermute(src << 2, __float_as_int(v)));
}

// packed 16x16 tiles, row stride 17
constexpr int TS = 17;
constexpr int TILE_ELEMS = 16 * TS;                 // 272
__device__ __forceinline__ int tri_index(int I, int J) { return I * (I + 1) / 2 + J; }


template <typename T>
constexpr int potf2_lds_bytes() { return (37 * TILE_ELEMS + NB) * (int)sizeof(T); }   // 36 tiles + current leaf inverse + pivots

// 16x16 Cholesky of a symmetric tile held in the MFMA accumulator layout (one wave).  Column j:
// pivot by readlane, rsqrt, scale column j (lanes with col == j), and one rank-1 MFMA update
// acc -= f f^T with f_c = S[j][c]/sqrt(d) for c > j (taken from row j by symmetry, zero elsewhere,
// so finished columns are never touched).  Returns the first failing column or -1.
// 1/sqrt(d): v_rsq_f64 (~26 bits) + one third-order correction step -- the refinement the precise library rsqrt
// performs, without its special-case selects (d is > 0 or already NaN here)
__device__ __forceinline__ double fast_rsqrt(double d) {
  const double y = __builtin_amdgcn_rsq(d);
  const double e = fma(-d * y, y, 1.0);
  return fma(y * e, fma(e, 0.375, 0.5), y);
}
__device__ __forceinline__ float fast_rsqrt(float d) { return rsqrt(d); }

// Cholesky of one symmetric 16x16 tile held in the MFMA C/D layout of one wave (both triangles present), and the
// inverse of its factor.  Column j: broadcast S[j][j] (readlane), inv = 1/sqrt, rank-1 update S -= f f^T on MFMA with
// f = row j * inv taken from the lanes that hold row j (symmetry: row j == column j, so no cross-lane transposition is
// needed).  The scaling of column j itself (L[:,j] = S[:,j] * inv) touches no later step -- updates only reach rows and
// columns > j -- and is applied once at the end.  The 16 steps are a serial dependency chain (the critical path of
// potf2).  The inverse rides in its shadow: V starts as I and takes the same eliminations, V -= f (inv * V[j,:]); row j
// of V is final after step j-1 and never touched again, and M = L^-1 = diag(1/L_jj) V.  Its MFMA is independent of the
// S chain and issues while the next pivot is being prepared.
// Four columns per step, no masks, results written as they become final.  The 4x4 pivot block
// P = S[j..j+3][j..j+3] reaches every lane through ten readlanes and its Cholesky factor (l10 l20 l30 l21 l31 l32, four
// inverse pivots) is computed redundantly by all lanes; rows j..j+3 of S and of V are gathered into every lane group
// (ds_bpermute), each lane forms the four elimination vectors f_t = (s_t - sum_{u<t} l_tu f_u) / pivot_t, lane group q feeds
// f_q as K slice q and ONE MFMA applies the four rank-1 updates (a second one takes V along).  f_q IS column j+q of L (its
// diagonal entry included) and h_q = inv_q (v_q - ...) IS row j+q of M = L^-1: lane group q stores them on the spot.  The
// updates therefore need no masks: entry (r, c) of S or V only sees a[r] and b[c], so what the un-masked vectors write into
// rows / columns that are already final is never read again (the tile's upper triangle holds leftovers; the final store of L
// zeroes it, M's upper triangle is exactly zero), and the per-column scaling pass at the end is gone.  A pivot <= 0 or NaN
// needs no branch either: v_rsq of it is NaN or inf, the Newton correction turns inf into NaN (0 * inf), NaN spreads --
// the failing column is read off the inverse pivots afterwards.  (The two-columns-per-step leaf with masks it replaced:
// 4084 cycles per leaf against ~3000, profiles/r02_potrf_chain.md.)
template <typename T>
__device__ __forceinline__ int leaf_cholesky4(typename Mma<T>::acc_t& acc, T* dt, T* sM, T* dinv_out, T* Wg, int64_t ldw, int lane) {
  typedef typename Mma<T>::acc_t acc_t;
  constexpr bool F64 = sizeof(T) == 8;
  const int l15 = lane & 15, lq = lane >> 4;
  acc_t vinv;
#pragma unroll
  for (int r = 0; r < 4; ++r) vinv[r] = (Mma<T>::crow(lane, r) == l15) ? (T)1 : (T)0;
#pragma unroll
  for (int j = 0; j < 16; j += 4) {
    // fp64: row = lq + 4 reg -> rows j+a sit in register j/4 of lane group a;  fp32: row = 4 lq + reg -> rows j+a sit in
    // register a of lane group j/4
    const int g0 = j >> 2;
    T s[4], v[4];
    T p00, p10, p11, p20, p21, p22, p30, p31, p32, p33;
    if constexpr (F64) {
      const T own = acc[g0], vown = vinv[g0];
      p00 = readlane_t(own, j);
      p10 = readlane_t(own, 16 + j); p11 = readlane_t(own, 16 + j + 1);
      p20 = readlane_t(own, 32 + j); p21 = readlane_t(own, 32 + j + 1); p22 = readlane_t(own, 32 + j + 2);
      p30 = readlane_t(own, 48 + j); p31 = readlane_t(own, 48 + j + 1); p32 = readlane_t(own, 48 + j + 2); p33 = readlane_t(own, 48 + j + 3);
#pragma unroll
      for (int a = 0; a < 4; ++a) { s[a] = lane_gather(own, 16 * a + l15); v[a] = lane_gather(vown, 16 * a + l15); }
    } else {
      const int base = 16 * g0;
      p00 = readlane_t(acc[0], base + j);
      p10 = readlane_t(acc[1], base + j); p11 = readlane_t(acc[1], base + j + 1);
      p20 = readlane_t(acc[2], base + j); p21 = readlane_t(acc[2], base + j + 1); p22 = readlane_t(acc[2], base + j + 2);
      p30 = readlane_t(acc[3], base + j); p31 = readlane_t(acc[3], base + j + 1); p32 = readlane_t(acc[3], base + j + 2); p33 = readlane_t(acc[3], base + j + 3);
#pragma unroll
      for (int a = 0; a < 4; ++a) { s[a] = lane_gather(acc[a], base + l15); v[a] = lane_gather(vinv[a], base + l15); }
    }
    // ---- Cholesky of the pivot block (all lanes, same values).  fp32: rsqrt(0) = inf has no Newton step to turn it into NaN
    auto inv_sqrt = [](T d) -> T {
      if constexpr (F64) return fast_rsqrt(d);
      else return d > (T)0 ? fast_rsqrt(d) : (T)NAN;
    };
    const T i0 = inv_sqrt(p00);
    const T l10 = p10 * i0, l20 = p20 * i0, l30 = p30 * i0;
    const T i1 = inv_sqrt(fma(-l10, l10, p11));
    const T l21 = fma(-l20, l10, p21) * i1, l31 = fma(-l30, l10, p31) * i1;
    const T i2 = inv_sqrt(fma(-l21, l21, fma(-l20, l20, p22)));
    const T l32 = fma(-l31, l21, fma(-l30, l20, p32)) * i2;
    const T i3 = inv_sqrt(fma(-l32, l32, fma(-l31, l31, fma(-l30, l30, p33))));
    dinv_out[j] = i0; dinv_out[j + 1] = i1; dinv_out[j + 2] = i2; dinv_out[j + 3] = i3;
    // ---- columns j..j+3 of L and the matching K slices of the update
    const T f0 = s[0] * i0;
    const T f1 = fma(-l10, f0, s[1]) * i1;
    const T f2 = fma(-l21, f1, fma(-l20, f0, s[2])) * i2;
    const T f3 = fma(-l32, f2, fma(-l31, f1, fma(-l30, f0, s[3]))) * i3;
    const T a = (lq == 0) ? f0 : (lq == 1 ? f1 : (lq == 2 ? f2 : f3));
    acc = Mma<T>::mma(-a, a, acc);
    __builtin_amdgcn_sched_barrier(0);
    // ---- rows j..j+3 of M = L^-1
    const T h0 = v[0] * i0;
    const T h1 = fma(-l10, h0, v[1]) * i1;
    const T h2 = fma(-l21, h1, fma(-l20, h0, v[2])) * i2;
    const T h3 = fma(-l32, h2, fma(-l31, h1, fma(-l30, h0, v[3]))) * i3;
    const T b = (lq == 0) ? h0 : (lq == 1 ? h1 : (lq == 2 ? h2 : h3));
    vinv = Mma<T>::mma(-a, b, vinv);
    dt[l15 * TS + j + lq] = a;                       // L[l15][j + lq]
    sM[(j + lq) * TS + l15] = b;                     // M[j + lq][l15]
    gst(Wg + (int64_t)(j + lq) * ldw + l15, b);      // leaf inverse (lower triangular; zeros above the diagonal)
    __builtin_amdgcn_sched_barrier(0);
  }
  // first column whose pivot was not positive (or NaN): its inverse pivot is NaN
  const T mine = dinv_out[l15];                      // (this wave's own stores: LDS operations of a wave stay in order)
  const unsigned long long badmask = __ballot(!(mine < (T)INFINITY)) & 0xFFFFull;
  return badmask ? (int)__builtin_ctzll(badmask) : -1;
}

