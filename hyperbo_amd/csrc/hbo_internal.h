// Internal declarations shared by the HIP translation units of libhbo (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/hbo.h"

// Pointers fetched from descriptor structs are generic ("flat") to the compiler; flat loads count on
// lgkmcnt as well as vmcnt, so every LDS wait would also drain the global prefetch.  These helpers
// force global_load / global_store (address space 1).
#define HBO_GLOBAL __attribute__((address_space(1)))
template <typename V> __device__ __forceinline__ V gld(const V* p) { return *(const HBO_GLOBAL V*)(p); }
template <typename V> __device__ __forceinline__ void gst(V* p, V v) { *(HBO_GLOBAL V*)(p) = v; }

#define HBO_TILE 128     // square tile / panel width of every blocked algorithm
#define HBO_N_COUNTERS 4096        // tile counters of one factorisation's persistent launches: [0, HBO_N_BULK_COUNTERS) the bulk trailing
#define HBO_N_BULK_COUNTERS 1024   // updates, the rest the inverse products that run beside the panel chain
#define HBO_LEAF 16      // MFMA tile edge (v_mfma_*_16x16x4)

// One GP sub-dataset (or one GPCache) as the kernels see it.  All matrices are row-major with
// leading dimension ld = npad (n rounded up to HBO_TILE); A has one extra tile-row (rows
// npad..npad+127) whose first row carries the residual  r = sum_a(y_a) - m*mu  so that the
// blocked Cholesky produces z = L^-1 r for free (see DESIGN.md "augmented row").
struct TaskDesc {
  void* A;           // (npad+128) x ld : Gram+jitter -> L (lower)
  void* W;           // npad x ld       : L^-1 (lower, zeros above)
  void* S;           // npad x ld       : scratch (trtri temp) -> K^-1 (lower tiles)
  void* wscr;        // ceil(npad/512) x ld : partial sums of W^T z (its own buffer: S may already hold a part of K^-1)
  const void* X;     // n x D inputs
  const void* F;     // n x fdim kernel features (== X when the kernel has no MLP)
  const void* Fm;    // n x fmean features feeding a linear mean (X or MLP output) or null
  const void* ysum;  // n : sum over the m columns of y
  void* svec;        // npad : s = K^-1 r (row-sum of kinvy)
  void* dF;          // n x fdim : d nll / d features (MLP kernels / linear_mlp mean)
  int n, npad, nblk, m;
  int64_t ld;
  int fdim, fmean;
  // generalised objective  f = c * sum_b |z_b|^2 + 2*lh * sum_i log L_ii + const  over `naug` augmented
  // rows  row_b = aug_src[b] + e_b * mu,  e_b = e_all + (b == naug-1 ? e_last : 0):
  //   NLL (objectives.py:144-156): 1 row (sum of y columns), e_last = -m, c = 1/2, lh = m^2/2
  //   EKL (objectives.py:29-101 with utils.partial_kl_mvn): rows yc_a/sqrt(m), last row -mu0, e_last = +1, c = 1, lh = 1
  //   factor path: m rows y_a, e_all = -1
  int naug;
  // divergence objectives with more than 127 aligned columns (m + 1 rows do not fit the augmented tile-row): only the LAST source row
  // (mu - mean_a y) rides in the tile (naug = 1, last_src = m); the m data rows are vectors of their own in svec columns 1..m --
  // EKL: alpha_b = W^T (W row_b) computed behind the inverse, EUC: the rows themselves (objective.hip: extra_rows) -- nvec = m + 1
  int nvec;          // outer-product vectors in svec (0: the naug vectors of the tile rows)
  int last_src;      // source row (of ysum / ydiv) behind tile row naug - 1; naug - 1 unless nvec
  double e_last, e_all, coef_c, coef_lh, coef_const;
  void* dmu;         // npad doubles: d f / d mu_i
  double* fnorm;     // 2 doubles: [0] |C0 - K1|_F, [1] |mu1 - mu0|  (EUC)
};

enum ObjectiveId { OBJ_NLL = 0, OBJ_EKL = 1, OBJ_EUC = 2 };

// Device-side copy of the (already warped) model hyper-parameters.
struct ModelDev {
  int kernel_id, mean_id, fdim, n_ls;
  double sv, noise, eps, constant, dot_sigma, dot_bias, linear_bias;
  double inv_ls[HBO_MAX_FEATURE_DIM];   // 1/lengthscale per feature (broadcast if n_ls==1)
  double lin_w[HBO_MAX_FEATURE_DIM];    // linear mean weights
};

enum GemmMode {
  GEMM_SYRK = 0,     // A[r,c] -= P[r,:] P[c,:]^T over panel columns (trailing / look-ahead update)
  GEMM_TRTRI_A = 1,  // S21 = L21 * W11
  GEMM_TRTRI_B = 2,  // W21 = -W22 * S21
  GEMM_LAUUM = 3,    // S = W^T W (lower tiles)
  GEMM_POST = 4,     // V = W * Kxq (column sums of squares and/or V itself)
  GEMM_VTV = 5,      // C -= V^T V over all npad rows (full posterior covariance: C holds Kqq on entry), tiles of mpad x mpad
  // the one-sweep inverse (sched.hip:sweep_advance), row group R = blocks [c_lo, c_hi) whose block columns of L are final;
  // S holds T (running products, rows below the front) and K^-1 (rows at and above it):
  GEMM_SWEEP_B = 6,  // W[i,j] = -sum_{k in R, k <= i} W[i,k] T[k,j]          i in R, j < c_lo
  GEMM_SWEEP_T = 7,  // T[i,j] (+)= sum_{k in R, k >= j} L[i,k] W[k,j]          i >= c_hi, j < c_hi   (first contribution: j in R)
  GEMM_SWEEP_C = 8,  // K^-1[i,j] (+)= sum_{k in R, k >= i} W[k,i]^T W[k,j]     j <= i < c_hi          (first contribution: i in R)
};

struct GemmArgs {
  const TaskDesc* tasks;
  int mode;
  int p0;        // SYRK: first panel tile-column of the K range;   TRTRI: half size s (tiles)
  int kt;        // SYRK: number of 128-wide K tiles;   TRTRI_B with a single group: tile rows launched
  int c_lo;      // SYRK: first updated tile column;   TRTRI_A: tile rows launched for the last group
  int c_hi;      // SYRK: end (exclusive) of updated tile columns;   TRTRI_A: index of the last group
  int aug;       // SYRK: 1 -> include the augmented tile-row
  int small_tiles;  // 1 -> 64x64 output tiles (SYRK / TRTRI / LAUUM; grid is given in 128-tile units)
  int persistent;   // >0 -> that many persistent workgroups loop over the tiles (SYRK; TRTRI on 128-tiles with a work_counter)
  int n_big;        // persistent SYRK on 128-tiles: tiles [0, n_big) of the linear order run as 128-tiles, the rest as four 64-tiles
                    // each (0 = all of them as 128-tiles): the last, partly filled round of a launch balances at a quarter of the grain
  int pgx, pgy;     // persistent TRTRI / SWEEP: the tile grid the workgroups walk (set by launch_gemm)
  int ptasks;       // persistent form over a batch: tiles x tasks are drawn from ONE counter (tile-major, so that equal-K tiles of
                    // all tasks are neighbours); 0: the task is blockIdx.z
  int* work_counter; // persistent SYRK: zero-initialised tile counter -> workgroups draw tiles dynamically (a faster
                     // workgroup takes more of them) instead of striding over them; null: static stride
  int dbg;          // HBO_GEMM_TIMING builds: 1 -> this launch records per-workgroup wall-clock stamps
  int* yield_flag;  // non-null (background launches: bulk update, overlapped inverse): per-CU table indexed by cu_token(); a
                    // workgroup sleeps at a K step while the entry of the CU it runs on is non-zero -- a panel-chain kernel
                    // (potf2, trsm, the chain's column updates) is running there and would otherwise share MFMA / LDS with it
  int* yield_mark;  // non-null (launches ON the panel chain): the same table; every workgroup counts itself in and out
  int grp_lo;       // TRTRI: first group (of size 2s blocks) handled by this launch
  void* B;       // POST: Kxq (npad x ldb)
  int64_t ldb;
  void* V;       // POST: optional V output (npad x ldb), may be null
  void* colsq;   // POST: partial column sums of squares [nblk][ldb], may be null
  // POST with few column tiles (kchunk > 0): the K range of a row tile is cut into chunks of kchunk 128-blocks, one workgroup per
  // (row tile, chunk); chunk ch of row tile i writes its partial product to V + (ch * npad + i * 128) * ldb (no colsq): split-K
  int kchunk;
  unsigned long long* tl;   // HBO_TIMELINE builds: [first start, last end] of this launch's workgroups (100 MHz wall clock), or null
};

#ifdef __HIPCC__
#define HBO_YIELD_TAB_ENTRIES 4224   // cu_token() < 4097
// identity of the CU a wave runs on: XCC_ID[3:0] and HW_ID[15:8] (cu, sh, se), never 0
__device__ __forceinline__ int cu_token() {
  const unsigned xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11));
  const unsigned hw = __builtin_amdgcn_s_getreg((4 << 0) | (8 << 6) | (7 << 11));   // HW_ID bits [15:8]
  return (int)(((xcc & 15u) << 8) | (hw & 255u)) + 1;
}
// x = h + m + l exactly, three bf16 numbers (round-to-nearest-even at every step; the remainders are exact in fp32): the operand
// form of the fp32 products on the bf16 matrix cores (post3.hip)
__device__ __forceinline__ void hbo_split3(float x, unsigned short& h, unsigned short& m, unsigned short& l) {
  const __bf16 bh = (__bf16)x;
  const float r1 = x - (float)bh;
  const __bf16 bm = (__bf16)r1;
  const float r2 = r1 - (float)bm;
  const __bf16 bl = (__bf16)r2;
  h = __builtin_bit_cast(unsigned short, bh); m = __builtin_bit_cast(unsigned short, bm); l = __builtin_bit_cast(unsigned short, bl);
}
// x s = h + l + r, |r| <= 2^-22 |x s|: two fp16 numbers (the caller scales by a power of two so that the operand's largest magnitude
// lands in [2^13, 2^14), hbo_h2_scale_for) -- the operand form of the fp32 products on the fp16 matrix cores (post2h.hip, post3.hip)
__device__ __forceinline__ void hbo_split2h(float x, unsigned short& h, unsigned short& l) {
  const _Float16 hh = (_Float16)x;
  const _Float16 ll = (_Float16)(x - (float)hh);
  h = __builtin_bit_cast(unsigned short, hh); l = __builtin_bit_cast(unsigned short, ll);
}
// power of two that maps `amax` into [2^13, 2^14)  (1 for amax = 0 / not finite: the planes then carry NaN / inf along; the
// exponent is clamped so that neither the scale nor a ratio of two scales leaves the fp32 range)
__device__ __forceinline__ float hbo_h2_scale_for(float amax) {
  if (!(amax > 0.f) || !(amax < INFINITY)) return 1.f;
  int e;
  (void)frexpf(amax, &e);   // amax = f 2^e, f in [0.5, 1)
  e = e < -40 ? -40 : e;
  return ldexpf(1.f, 14 - e);
}
// a panel-chain workgroup announces itself on its CU (background GEMM workgroups there pause, see GemmArgs::yield_flag)
__device__ __forceinline__ int yield_enter(int* tab) {
  const int tok = cu_token();
  __hip_atomic_fetch_add(tab + tok, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return tok;
}
__device__ __forceinline__ void yield_leave(int* tab, int tok) {
  __hip_atomic_fetch_add(tab + tok, -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// debug builds (-DHBO_TIMELINE, tools/timeline.py): every workgroup of a tagged launch folds its start / end into the launch's slot pair
__device__ __forceinline__ void tl_begin(unsigned long long* tl) {
#ifdef HBO_TIMELINE
  if (tl && threadIdx.x == 0) atomicMin(tl, (unsigned long long)wall_clock64());
#endif
}
__device__ __forceinline__ void tl_end(unsigned long long* tl) {
#ifdef HBO_TIMELINE
  if (tl && threadIdx.x == 0) atomicMax(tl + 1, (unsigned long long)wall_clock64());
#endif
}
#endif

// hipFuncSetAttribute is per device: a launcher sets its kernels' dynamic-LDS limit the first time it runs on EACH device of the
// process (one bit per device ordinal; a second context on another GPU would otherwise launch with the 64 KB default and fail)
static inline bool hbo_first_use_on_device(unsigned long long& seen) {
  int d = 0;
  if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= 64) return true;
  if ((seen >> d) & 1ull) return false;
  seen |= 1ull << d;
  return true;
}

// ---- launchers (defined in the .hip files) ---------------------------------------------
void launch_gemm(int dtype, const GemmArgs& a, dim3 grid, hipStream_t st);

// fp32: the panel solve also writes the solved panel as three bf16 planes (the operand of the bf16x3 trailing updates, post3.hip):
// element (row, k) of panel block column kb_off.. -> ((row / 128 * nkb + kb_off + k / 16) * 3 + plane) * 2048 + (row % 128) * 16 + k % 16
struct SplitOut { unsigned short* xp; int64_t task_stride; int nkb, kb_off; };
void launch_potf2(int dtype, const TaskDesc* tasks, int ntasks, int p, int* info, hipStream_t st, int* yield_flag = nullptr,
                  unsigned long long* tl = nullptr);
void launch_trsm(int dtype, const TaskDesc* tasks, int ntasks, int p, int max_nblk, hipStream_t st, int* yield_tab = nullptr,
                 const SplitOut* so = nullptr, unsigned long long* tl = nullptr);
// HBO_TIMELINE builds: the slot pair of the next tagged launch (sched.hip), null when no recording is on
unsigned long long* tl_slot(const char* name, int p);
// inverses of the diagonal blocks p in [p_lo, p_hi)
void launch_trtri_diag(int dtype, const TaskDesc* tasks, int ntasks, int p_lo, int p_hi, hipStream_t st);

struct GramArgs {
  const TaskDesc* tasks;   // batched symmetric mode (tasks != null): out = tasks[z].A, x = tasks[z].F
  const void* x1; const void* x2; void* out;  // direct mode
  int64_t n1, n2, ldo;
  int n1pad, n2pad;        // direct padded mode: extents up to which zeros are written
  int fdim;
  int symmetric;           // lower tiles only + (noise+eps) on the diagonal + identity padding
  int padded;              // out has a padded ld/extent (16-byte vector stores, zero fill)
  int kernel_id;           // covariance of the model behind the ModelDev pointer (selects the kernel instantiation)
  int direct_form;         // fp32: keep the VALU kernel's sum (a - b)^2 even where gram_mfma_kernel would apply (see launch_gram_t)
  int model_stride;        // batched mode: task z reads the model at md + z * model_stride (0: one model for the batch; 1: one
                           // ModelDev per task -- S hyper-parameter samples of one model family factorised as one batch)
};
void gram_set_mfma_min_features(int f);   // hbo_tune gram_mfma
void launch_gram(int dtype, const GramArgs& a, const ModelDev* model, dim3 grid, hipStream_t st);
void launch_kdiag(int dtype, const void* f, int64_t n, int fdim, const ModelDev* model, void* out,
                  hipStream_t st);
void launch_dense_tanh(int dtype, const void* in, const void* w, const void* b, void* out, int64_t n,
                       int fin, int fout, hipStream_t st);
void launch_mean(int dtype, const void* fm, int64_t n, int fmean, const ModelDev* model, void* mu,
                 hipStream_t st);
void launch_aug_rows(int dtype, const TaskDesc* tasks, int ntasks, int max_npad, const ModelDev* md, hipStream_t st, int model_stride = 0);
void launch_poison(int dtype, const TaskDesc* tasks, int ntasks, int max_npad, hipStream_t st);   // hbo_tune "poison": NaN into everything about to be recomputed
void launch_nll_reduce(int dtype, const TaskDesc* tasks, int ntasks, const int* info, double* out,
                       hipStream_t st);
// s = W^T z_a (z_a = augmented row a of A) -> tasks[t].svec[out_col*out_ld + j]; uses wscr as scratch
// xover / oover (single task only): explicit input vector [npad] / output vector [npad]
void launch_wt_z(int dtype, const TaskDesc* tasks, int ntasks, int max_nblk, int aug_row, int out_col,
                 int out_ld, hipStream_t st, const void* xover = nullptr, void* oover = nullptr);
int grad_nacc(int kernel_id, int fdim);   // accumulators per tile (incl. the Frobenius slot)
void launch_grad_contract(int dtype, const TaskDesc* tasks, int ntasks, int max_nblk, const ModelDev* md,
                          int kernel_id, int fdim, int obj, double* partials, int64_t stride_task, hipStream_t st);
// d f / d mu_i per task (tasks[t].dmu) -- needs svec (NLL/EKL) or the data rows (EUC)
void launch_dmu(int dtype, const TaskDesc* tasks, int ntasks, int obj, hipStream_t st);
// value_out (nullable, EUC): per-task objective value |mu1-mu0| + |C0-K1|_F
void launch_grad_finalize(int dtype, const TaskDesc* tasks, int ntasks, const ModelDev* md, int kernel_id,
                          int fdim, int obj, const double* partials, int64_t stride_task, double* out,
                          int out_stride, double* value_out, hipStream_t st, double* pre = nullptr, int max_nblk = 0);
constexpr int HBO_GRAD_PRE_ROWS = 32;   // rows of the pre-reduction buffer per task (`pre`: ntasks x 32 x nacc doubles)
void launch_scale_dF(const TaskDesc* tasks, int ntasks, int64_t max_n, int fdim, hipStream_t st);
void launch_grad_feat(int dtype, const TaskDesc* tasks, int ntasks, int max_nblk, const ModelDev* md, int kernel_id,
                      int fdim, int obj, hipStream_t st);
void launch_grad_feat_mean(int dtype, const TaskDesc* tasks, int ntasks, int64_t max_n, const ModelDev* md,
                           int fdim, hipStream_t st);
void launch_dense_bwd(int dtype, const void* in, const void* out, const void* w, double* dout, double* din,
                      double* dW, double* db, int64_t n, int fin, int fout, hipStream_t st);
// NLL (+ gradient block in grad_finalize's layout) of a batch whose tasks all have n <= 128: one workgroup per task (small.hip).
// write_back: also store K^-1 (full n x n square in S), s = K^-1 r (svec) and d f / d mu (dmu) for the feature-gradient kernels of
// an MLP model (grad_feat_kernel, grad_feat_mean_kernel), which follow as launches of their own
int small_eval_lds(int dtype);
void launch_small_eval(int dtype, const TaskDesc* tasks, int ntasks, const ModelDev* md, int kernel_id, int fdim, int* info,
                       double* nll_out, double* grad_out, int out_stride, int write_back, hipStream_t st);
// MLP basis of a whole batch, one launch per layer (mlp.hip): per-task pointers
struct MlpTaskDev {
  const void* x;                        // n x D inputs
  void* acts[HBO_MAX_MLP_LAYERS];       // n x f_l activations
  double* dF; double* dtmp;             // backward: gradient w.r.t. a layer's output, ping-pong
  int64_t n;
};
void launch_mlp_forward_batch(int dtype, const MlpTaskDev* mt, int ntasks, int64_t max_n, int layer, const void* w, const void* b, int fin,
                              int fout, hipStream_t st);
void launch_mlp_zero_dF_batch(const MlpTaskDev* mt, int ntasks, int64_t max_n, int flast, hipStream_t st);
// one layer of the backward pass for every task: dz in place on the current buffer (dF when cur_is_dF, else dtmp), dW / db summed
// over rows and tasks (fp64 atomics), d input into the other buffer when want_din
void launch_dense_bwd_batch(int dtype, const MlpTaskDev* mt, int ntasks, int64_t max_n, int layer, int cur_is_dF, const void* w, double* dW,
                            double* db, int fin, int fout, int want_din, hipStream_t st);
struct PostArgs {
  const void* Kxq; int64_t ldq; int npad; int n; int nblk;   // cross Gram (npad x ldq)
  const void* alpha;    // kinvy [npad] (first column)
  const void* colsq;    // partial [nblk][ldq]
  void* mupart;         // scratch [nblk][ldq] for the row-block partial sums of Kxq^T alpha (null: one thread per query walks all rows)
  const void* kdiag;    // prior variance at the queries [M]
  const void* muq;      // prior mean at the queries [M]
  void* mu_out; void* var_out; void* acq_out;
  int64_t M;
  int acq_id; double param, add_noise, scale;
};
void launch_post_epilogue(int dtype, const PostArgs& a, hipStream_t st);
// colsq[i][col] = sum over the 128 rows of row block i of (sum_ch Vpart[ch][row][col])^2, chunks ch < ceil((i + 1) / kchunk)
void launch_post_colsq_split(int dtype, const void* vpart, int npad, int64_t ldq, int mpad, int nblk, int kchunk, void* colsq, hipStream_t st);

// ---- fp32 posterior product on the bf16 matrix cores (post3.hip) --------------------------------
struct Post3Args {
  const unsigned short* Wp;   // W = L^-1 split into bf16 panel blocks (layout: post3.hip), npad rows, k up to npad
  const unsigned short* Kp;   // Kxq^T split the same way: rows = candidates of the chunk, k = training points
  int nkb;                    // npad / 16: blocks of k per 128-row tile (both operands)
  float* colsq; int64_t ldc;  // [nblk][ldc] per-row-block column sums of V^2 (may be null)
  float* V; int64_t ldv;      // optional V output (npad x ldv)
  int nblk;
  int col_tiles;              // (set by launch_post3)
  int* work_counter;          // non-null: a resident grid draws the (row tile, column tile) pairs from this zeroed counter, long rows first
};
// fp32 posterior product on the fp16 matrix cores from two-way splits (post2h.hip): operands scaled by powers of two into fp16's range
struct Post2hArgs {
  const unsigned short* Wp;   // W = L^-1 split into fp16 panel blocks (two planes; layout: post2h.hip), scaled by scale_for(*wmax_bits)
  const unsigned short* Kp;   // Kxq^T split the same way, scaled by kscale
  int nkb;
  float* colsq; int64_t ldc;  // [nblk][ldc] per-row-block column sums of V^2
  int nblk;
  int col_tiles;              // (set by launch_post2h)
  int* work_counter;          // as Post3Args
  const unsigned int* wmax_bits;   // device: bits of max |W| (launch_absmax_lower)
  float kscale;
};
void launch_absmax_lower(const float* in, int64_t ld, int nblk, unsigned int* out, hipStream_t st);
void launch_split2h_rows(const float* in, int64_t ld, int row_tiles, unsigned short* out, int nkb, const unsigned int* amax_bits, hipStream_t st);
void launch_split2h_transpose(const float* in, int64_t ld, int krows, int jcols, unsigned short* out, int nkb, float scale, hipStream_t st);
float post2h_scale_for(double bound);
void launch_post2h(const Post2hArgs& a, int col_tiles, hipStream_t st);
// fp32 trailing updates of the blocked Cholesky on the bf16 matrix cores (post3.hip: split3_panel_kernel, syrk3_kernel)
struct Syrk3Args {
  const TaskDesc* tasks;
  unsigned short* Xp;        // split copy of the current group's panels: per task, row tile R (0..nblk, nblk = augmented tile-row),
  int64_t task_stride;       //   k block KB (0..nkb), three planes of 128 x 16 bf16 each; task_stride elements per task
  int nkb;                   // k blocks per row tile in the buffer (16 panel columns each)
  // split: columns [kcol0, kcol0 + 16 nk_split) of A go to blocks [kb_off, kb_off + nk_split) of the row tiles r_lo...
  int kcol0, nk_split, r_lo;
  // product: C[r, c] -= X[r, kb_off .. kb_off + nk) X[c, ...]^T for the tiles c in [c_lo, c_hi), r in [c, nblk]
  int kb_off, nk, c_lo, c_hi;
  // persistent form: `persistent` workgroups draw the tiles from *work_counter (zeroed by the caller) -- fewer workgroups than the
  // machine holds, so that the panel kernels always find a CU with room
  int persistent; int* work_counter;
  int* yield_flag;   // background launch (the bulk update): pause at a pipeline step while a panel-chain workgroup runs on this CU
  int* yield_mark;   // launch ON the panel chain (F1, column updates): count the workgroup into the same per-CU table
  // mode 0: the trailing update above.  Modes 1 / 2: the products of the block-recursive inverse (sched.hip:trtri_level) on one
  // level of s blocks for `ngrp` groups, operands split by split3_block / split3_block_t into Xp (A operand) and Yp (B operand):
  //   1 (TRTRI_A): S21[it, jt] =  sum_{k >= jt} L21[it, k] W11[k, jt]     A = L21 rows, B = W11^T rows, K blocks [8 jt, 8 s)
  //   3 (LAUUM): S[i, jt] = sum_{k >= i} W[k, i]^T W[k, jt] for the lower tiles, Xp = split transpose of W (nkb = 8 nblk)
  //   2 (TRTRI_B): W21[it, jt] = -sum_{k <= it} W22[it, k] S21[k, jt]     A = W22 rows, B = S21^T rows, K blocks [0, 8 (it + 1))
  // tile index = ((grp * vt + it) * s + jt) in launch order (longest K first); per group the operand tiles are contiguous:
  // A tile it of group g at ((g * s + it) * nkb), B tile jt at ((g * s + jt) * nkb), nkb = 8 s blocks.
  int mode, s, grp_lo, ngrp, vlast;   // vlast: tile rows of the LAST group of the launch (a cut lower half), s for the others
  unsigned short* Yp;
  // h2 = 1: the f16x2 form -- operands as TWO fp16 planes of values scaled by a power of two (hbo_split2h), three MFMAs per pair of
  // fragments instead of six, the result scaled back in the epilogue.  sx / sy: scales of the A / B operand known to the host
  // (the factor's entries: |L_ij| <= sqrt(max_i A_ii)); sx_bits / sy_bits: device words that override them -- the bits of a measured
  // maximum, from which split and product derive the same power of two (hbo_h2_scale_for).  Mode 0: the augmented tile-row has no a-priori bound (z = L^-1 r): its split takes
  // the maximum of every 16 rows x 64 panel columns and writes the scale to aug_scale[task][k block / 4][row / 16]; the product
  // rescales its accumulators of that tile-row where the scale changes (exact: powers of two).
  int h2;
  float sx, sy;
  const unsigned int* sx_bits; const unsigned int* sy_bits;
  float* aug_scale; int64_t aug_stride;
  unsigned int* max_out;   // != null: atomicMax of the bits of max |result| over the launch's tiles (scale of a later split)
};
// split of a (row tiles x 128 nkb/8 ...) fp32 block into panel blocks: rows of `in` are the operand rows (k along the row) --
// or, transposed, columns of `in` are the operand rows (k along the column).  grid z = group, stepping `in` by gstep elements and
// `out` by gstride elements
struct Split3Block { const float* in; int64_t ld, gstep; unsigned short* out; int64_t gstride; int row_tiles, nkb, tri;
                     int last_rows, last_krows;   // the LAST group: operand row tiles / k rows that exist (a cut lower half)
                     // h2: two fp16 planes scaled by `scale` (host-known) -- or by the scale of a measured maximum: *scale_bits,
                     // or (max_out) taken by a first pass over the same elements into *max_out (zeroed by the caller, or extended)
                     int h2; float scale; const unsigned int* scale_bits; unsigned int* max_out; };
void launch_split3_block(const Split3Block& a, int ngrp, bool transposed, hipStream_t st);
void launch_split3_panel(const Syrk3Args& a, int row_tiles, int ntasks, hipStream_t st);
void launch_syrk3(const Syrk3Args& a, int ntiles, int ntasks, hipStream_t st);
void launch_split3_rows(const float* in, int64_t ld, int row_tiles, unsigned short* out, int nkb, hipStream_t st);
void launch_split3_transpose(const float* in, int64_t ld, int krows, int jcols, unsigned short* out, int nkb, hipStream_t st, int lower_only = 0);
void launch_split2h_transpose_measured(const float* in, int64_t ld, int krows, int jcols, unsigned short* out, int nkb, unsigned int* amax_bits, hipStream_t st, int lower_only);
void launch_post3(const Post3Args& a, int col_tiles, hipStream_t st);

struct AcqGradArgs {
  const void* Fq; const void* F; int fdim; int64_t n; int npad;   // kernel features: queries [M][fdim], training [n][fdim]
  const void* Kq; const void* L; const void* B;   // [M][npad]: k(x_q, X), W k, W^T W k  (null when n == 0)
  const void* alpha; const void* kdiag; const void* muq;
  int acq_id; double param, add_noise, scale;
  void* acq_out;     // [M] model dtype
  double* gfeat;     // [M][fdim] d acq / d kernel features
  double* dmu;       // [M] d acq / d mu
  int64_t M;
};
void launch_acq_grad(int dtype, const AcqGradArgs& a, const ModelDev* md, hipStream_t st);
void launch_acq_grad_mean(const double* dmu, const ModelDev* md, int64_t M, int fm, double* out, int accumulate,
                          hipStream_t st);
void launch_add_inplace(double* dst, const double* src, int64_t count, hipStream_t st);
void launch_extract_lower(int dtype, const void* A, int64_t ld, int64_t n, void* out, hipStream_t st);
void launch_symmetrize_from_lower(int dtype, const void* S, int64_t ld, int64_t n, void* out, hipStream_t st);
void launch_fill_spd(int dtype, const void* a_dense, int64_t n, void* A, int64_t ld, int npad, hipStream_t st);
void launch_set_aug(int dtype, const void* b, int64_t n, int m, void* A, int64_t ld, int npad, hipStream_t st);
// divergence objectives beyond 127 aligned columns: dst[b][j] = j < n ? src[b][j] : 0 for `count` rows of n -> npad elements, and
// value[0] += coef * sum_b sum_j z[b][j]^2
void launch_expand_rows(int dtype, const void* src, int64_t n, int npad, void* dst, int count, hipStream_t st);
void launch_add_sumsq(int dtype, const void* z, int npad, int count, double coef, double* value, hipStream_t st);
void launch_tri_matvec(int dtype, const void* W, int64_t ld, int npad, const void* x, int64_t xld, int m,
                       int trans, void* out, int64_t old, hipStream_t st);
