/* hbo_tune.h -- measurement hooks of libhbo.  NOT part of the drop-in boundary (include/hbo.h): nothing a caller of the
 * hyperbo.gp_utils surface needs.  The A/B tools under tools/ and the scheduling sweep of tests/test_gpu_fuzz.py use it to
 * vary where and when the same kernels run; no knob changes a result, and every non-default value was measured equal or worse
 * (profiles/r01_potrf_chain.md, r02_potrf_chain.md, r03_dag.md, r04_chain_and_sweep.md, r04_gemm_pipeline.md).  Names fall through to hbo_set_option.
 *
 *   overlap_trtri 0/1      inverse walks the block tree while the factorisation runs
 *   cu_yield      0..2     background GEMM workgroups pause while a panel-chain workgroup runs on their CU (1: potf2 only)
 *   persist_free  -1..200  CUs the persistent bulk update leaves with one workgroup (-1 = auto: 32)
 *   trtri_at      0..63    the overlapped inverse starts after this many 64ths of the panels (0 = auto: 5/8)
 *   trtri_free    0..200   CUs the co-running inverse products leave with one workgroup
 *   post_bf16x3 / syrk_bf16x3 / trtri_bf16x3 / lauum_bf16x3  0/1   the four parts of option bf16x3 separately
 *   trtri3_min_s  >=1      lowest level (in blocks) of the inverse that runs on the bf16 cores
 *   syrk3_col / syrk3_sep / syrk3_free       fp32 trailing updates: column updates on the bf16 cores too / panels split by
 *                          their own kernel instead of inside the panel solve / CUs the bulk update leaves with one workgroup
 *   sweep         0..2     one-sweep inverse (W = L^-1 and K^-1 = W^T W row group by row group behind the panel chain): 0 never,
 *                          1 where measured faster (default: batches with look-ahead, one fp64 matrix of 17-48 blocks;
 *                          profiles/r04_chain_and_sweep.md, r04_gemm_pipeline.md), 2 wherever look-ahead is on
 *   sweep_qs      0..16    its row-group size in 128-blocks, a power of two (0 = auto: 4; 8 for one matrix above 28 blocks)
 *   sweep_free    1..200   CUs the sweep's persistent launches beside the chain leave with one workgroup (24; at most trtri_free)
 *   sweep_side    0/1      one matrix on 128-tiles: the sweep's K^-1 updates on a stream of their own (default 1: N = 6144 6.07 -> 5.71 ms)
 *   lauum_persist 0..128   one large matrix: K^-1 = W^T W and the big levels of the inverse behind the factorisation as resident grids drawing
 *                          their tiles from a counter (default 1; N = 8192 10.64 -> 10.51 ms; n > 1: K^-1 leaves n CUs free instead of 16)
 *   split_f1      0..2     the update of the next group's block columns (F1): only the next column on the panel stream, the later
 *                          ones on a third stream -- 0 never, 1 batches (default: 64 tasks 14.32 -> 14.12 ms), 2 always
 *   sweep_big     >=0      sweep launches of small / batched shapes with at least this many 128-tiles (x tasks) use 128-tiles (4000)
 *   post_f16x2    0/1      fp32 posterior product of the stationary covariances on the fp16 matrix cores from two-way splits (post2h.hip;
 *                          default 1; 0 = bf16x3's exact three-way split)
 *   group_inner   -1..16   two-level panel groups: the chain's left-looking column updates stay inside inner groups of this many panels, one
 *                          update per inner boundary brings the rest of the (outer, potrf_group) group up to date; 0 = one level, -1 = auto
 *                          (8 inside groups of 16 for one fp32 matrix above 96 blocks: cfg 3's factor 18.5 -> 18.1 ms)
 *   chol_f16x2    0/1      fp32 factorisations of the stationary covariances (the objective / factor paths, which know max A_ii = signal variance +
 *                          noise + jitter): trailing updates, the inverse's bf16-core levels and K^-1 = W^T W from two-way fp16 splits
 *                          (post3.hip: syrk3_kernel<true>; default 1; 0 = the exact three-way bf16 splits, which hbo_spd_* and the
 *                          dot-product kernel always use)
 *   post_serial   0/1      streamed posterior: features + cross Gram of chunk i+1 on the SAME stream as the product of chunk i
 *                          (nothing overlaps: the stage times of hbo_profile are then each kernel's isolated time; bench.py cfg3)
 *   small_fused   0/1      batches whose tasks all have n <= 128: the single-workgroup evaluation (small.hip; default 1)
 *   poison        0/1      tests: an evaluation (objective / factor paths on the blocked pipeline) first fills what it is about to recompute --
 *                          the lower triangles of A and W, all of S, alpha, d f / d mu -- with NaN, so that a launch that skips work shows up
 *                          as NaN instead of hiding behind an earlier evaluation's identical numbers in the same pooled buffers; tests/conftest.py
 *                          turns it on for the whole GPU tier (default 0)
 *   fault_shard   0..2     ONE-SHOT fault injection into the next hbo_objective_sharded call of this context (tests of the failure
 *                          paths): 1 = the rank's local part counts as failed -> it joins the all-reduce with NaN in every slot;
 *                          2 = and it cannot produce that buffer either -> ncclCommAbort, the peers' all-reduce fails, later sharded
 *                          calls return HBO_ERR_COMM until hbo_comm_init
 *   f2_split      0..2     round 6: the bulk trailing update's leading block columns -- at least the next group's, about one resident round of tiles --
 *                          as a launch of their own with the event the next F1 waits for behind THAT launch (1: one matrix, 2: batches too).
 *                          Identical values; measured neutral (N = 8192 10.51 -> 10.57 ms: profiles/r06_f2_split.md); default 0
 *   gram_mfma     0..4096  PROCESS-WIDE: fp32 Gram matrices of the stationary covariances with at least this many features take gram_mfma_kernel
 *                          (u = |a|^2 + |b|^2 - 2 a.b, the dot product from exact three-way bf16 splits on the matrix cores; default 32;
 *                          0 = always the direct form sum (a - b)^2; profiles/r06_gram_mfma.md)
 *   batch_bg      -1..2    batches: the sweep's launches beside the chain as plain grids (0), persistent over tiles x tasks
 *                          from one counter (1), and also polling the per-CU yield table the chain's kernels then fill (2);
 *                          -1 (default): 1 up to 8 tasks, 0 above (8 tasks 2.52 -> 2.44 ms, 64 tasks 14.13 / 14.31) */
#ifndef HBO_TUNE_H_
#define HBO_TUNE_H_
#include "hbo.h"
#ifdef __cplusplus
extern "C" {
#endif
int hbo_tune(hbo_ctx* ctx, const char* name, int64_t value);
/* sustained fp64 MFMA rate of the device, measured now by ~ms milliseconds of back-to-back v_mfma_f64_16x16x4_f64 on every SIMD
 * (TFLOP/s): the roofline denominator bench.py reports beside the datasheet figure */
int hbo_mfma_peak_probe(hbo_ctx* ctx, double ms, double* tflops_out);
#ifdef __cplusplus
}
#endif
#endif /* HBO_TUNE_H_ */
