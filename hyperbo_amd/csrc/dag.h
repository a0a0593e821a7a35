// Resident tile-task executor of the blocked Cholesky (dag.hip): ONE persistent kernel runs every GEMM-shaped tile of the
// factorisation phase -- the in-group column updates, the next group's block columns, the bulk trailing update and the
// products of the block-recursive inverse that become computable while the panel chain still runs -- as tasks drawn from
// priority queues, each gated by dependency counters; the narrow panel kernels (potf2, the panel solve, the inverses of the
// diagonal blocks) stay host-launched on the panel stream, poll the same counters before they touch a tile and count
// themselves done after it.  Replaces the launch-per-step schedule of sched.hip:run_potrf for the same arithmetic
// (hyperbo/basics/linalg.py:29-33: cholesky of K + (noise + eps) I; NaN on a non-positive pivot).
#pragma once
#include "hbo_internal.h"

enum DagMode : int {
  DAG_SYRK = 0,      // C[r,c] -= P[r, p0..p0+kt) P[c, p0..p0+kt)^T   (r, c in tm-units)
  DAG_TRTRI_A = 1,   // S21 = L21 W11 of the tree node (s = p0 blocks, group = kt), tile (it = r, jt = c)
  DAG_TRTRI_B = 2,   // W21 = -W22 S21
};

struct DagTask {
  unsigned char mode, tm64, ndep, pad0;   // tm64: 1 -> 64x64 tile, 0 -> 128x128
  short mat;                              // matrix (task of the batch)
  short r, c, p0, kt;
  int dep_idx[3]; int dep_thr[3];         // counter index (inside the matrix's counter block) >= threshold
  int done_idx, done_inc;                 // counter bumped after the tile is visible device-wide
};

struct DagSeg { int start, len; };
// debug stamps per panel p (index p * 16 + k), wall_clock64 ticks (10 ns):
//  0 min: first panel-solve workgroup of p done   1 max: last one done
//  2 min: first task of queue-0 segment p drawn    3 min: first one started   4 max: last one published
//  5 min: potf2(p) entered  6 min: potf2(p) passed its wait  7 max: potf2(p) end
//  8 min: first panel-solve workgroup of p passed its wait  9 max: last one passed
enum { DAG_STAMPS_PER_PANEL = 16 };        // a run of tasks that becomes available together; its cursor is ctr[off_cursor + seg]

constexpr int DAG_NQ = 3;                 // queues in priority order: 0 chain-critical updates, 1 bulk updates, 2 inverse products
constexpr int DAG_VER_UNIT = 4;           // tile-version counters advance by 4 per applied update (a 64x64 quadrant adds 1)

// Per-matrix counter block (indices relative to DagDev::off_mat + mat * stride), M = max block count of the batch:
//   VER  [r * M + c]            updates applied to tile (r, c) x DAG_VER_UNIT    (r <= M: the augmented tile-row is row nblk)
//   ROW  [p * (M+1) + rb]       panel-solve workgroups of panel p done with row block rb (2 = final)
//   DIAG [p]                    workgroups of the diagonal-block inverse of block p done (2 = final)
//   NA / NB [li * NG + grp]     A- / B-tiles of tree node (level li, group grp) done
struct DagLayout {
  int M, NG, stride, off_ver, off_row, off_diag, off_na, off_nb;
  __host__ __device__ int ver(int r, int c) const { return off_ver + r * M + c; }
  __host__ __device__ int row(int p, int rb) const { return off_row + p * (M + 1) + rb; }
  __host__ __device__ int diag(int p) const { return off_diag + p; }
  __host__ __device__ int na(int li, int g) const { return off_na + li * NG + g; }
  __host__ __device__ int nb(int li, int g) const { return off_nb + li * NG + g; }
};
static inline DagLayout dag_layout(int M) {
  DagLayout l; l.M = M; l.NG = M / 2 + 1;
  l.off_ver = 0; l.off_row = l.off_ver + (M + 1) * M; l.off_diag = l.off_row + M * (M + 1);
  l.off_na = l.off_diag + M; l.off_nb = l.off_na + 12 * l.NG; l.stride = l.off_nb + 12 * l.NG;
  return l;
}

// global counter words
enum { DAG_ABORT = 0, DAG_NSEEN = 1, DAG_STAT_TASKS = 2, DAG_STAT_IDLE = 3, DAG_HINT = 8 /* + queue */, DAG_SE_SEEN = 32 /* + (XCC_ID, SE_ID, SH_ID) = (cu_token - 1) >> 4 */, DAG_FIXED = 32 + 256 };
constexpr int DAG_CU_TAB = 4224;                 // == HBO_YIELD_TAB_ENTRIES: cu_token() < 4097   // entries per CU table (registration count, decision)

struct DagDev {
  const TaskDesc* mats;
  const DagTask* tasks;
  const DagSeg* segs;
  int* ctr;
  int q_first[DAG_NQ + 1];   // segments of queue q: [q_first[q], q_first[q+1])
  int off_cursor;            // ctr index of the first segment cursor
  int off_reg, off_dec;      // per-CU tables: workgroups registered / decision (1 work, 2 leave the CU to the panel chain)
  int off_mat;               // first per-matrix counter block
  int stride;                // words per matrix
  int reserve;               // CUs PER SHADER ENGINE (4 per XCD) left to the panel chain: the dispatcher deals workgroups to the XCDs and,
                             // inside an XCD, to its shader engines in a fixed rotation whatever the load -- a workgroup whose turn falls
                             // on an engine without room waits there (measured: with the free CUs on some engines only, half of a panel
                             // solve's workgroups never started)
  int q_group;               // panels per group
  int idle_sleep;            // microseconds (about) an idle scheduler sleeps between two looks at the queues
  int spin_us;               // how long a workgroup keeps waiting for a drawn chain-critical task before it serves other queues
  unsigned long long* stamps; // HBO_DAG_DEBUG builds: wall-clock stamps per panel (see tools/dag_stamps.py), else null
  int dbg_flags;             // experiments: 1 = no acquire before a tile, 2 = no release after it (timing only, results may be stale)
  long long timeout_ticks;   // wall-clock bound (100 MHz) on the whole kernel: abort flag and exit
};

// What a panel-chain kernel polls before it touches its tiles and bumps when it is done (ctr == null: launch-ordered schedule).
struct ChainSync {
  int* ctr;             // global counter array
  int off_mat, stride;  // per-matrix blocks
  DagLayout lay;
  int need;             // potf2 / panel solve of panel p: VER threshold of column p's tiles
  unsigned long long* stamps;
  long long timeout_ticks;
};
