// Resident tile-task schedule of the blocked right-looking Cholesky on gfx950 (see dag.h for the idea and the counters).
//
//   dag_worker_kernel : two 256-thread workgroups per CU on all but `reserve` CUs (a workgroup that finds itself on one of
//                       the first `reserve` CUs of its shader engine to register exits: those CUs belong to the panel kernels for the whole
//                       phase, no co-residence with a 256-VGPR GEMM workgroup, no yield table).  Lane 0 of a workgroup is
//                       its scheduler: queues in priority order, per-segment cursors (a draw never spills into a segment
//                       that is not open yet), dependency counters polled with relaxed agent-scope loads, one acquire
//                       before the tile, one release + counter bump after it.  A drawn task whose dependencies are still
//                       short is held while the workgroup serves lower queues -- every task ahead of a held one has been
//                       drawn by somebody, so the schedule cannot deadlock -- except that a chain-critical task is waited
//                       for on the spot for `spin_us` first (its inputs are microseconds away).
//   build_dag         : host, once per batch shape: the task lists in the order the launch schedule would have run them.
//   run_potrf_dag     : host: counters zeroed, worker kernel on the main stream, potf2 / panel solve / diagonal inverses
//                       on the panel streams with ChainSync.
//
// Same arithmetic as sched.hip:run_potrf (hyperbo/basics/linalg.py:29-33); which tiles a GEMM workgroup runs and when is
// all that changes.
#define HBO_DEVICE_ONLY
#include "gemm.hip"
#undef HBO_DEVICE_ONLY
#include "dag_sync.h"
#include "sched.h"

#include <algorithm>
#include <array>
#include <string.h>
#include <map>
#include <vector>
#include <thread>
#include <chrono>
#include <stdlib.h>

#ifdef HBO_DAG_DEBUG
static unsigned long long* g_dag_stamps = nullptr;
#endif
namespace {

constexpr int DAG_LDS_BYTES = GEMM_LDS_BYTES + 64;   // the tile buffers + the scheduler's mailbox

template <typename T, int TM>
__device__ __forceinline__ void decode_dag_task(const DagTask& k, const TaskDesc& t, TileJob<T>& j) {
  constexpr int BKE = 128 / sizeof(T);
  constexpr int U = HBO_TILE / TM;
  const int64_t ld = t.ld;
  j.colsq = nullptr; j.yield_flag = nullptr;
  j.lda = j.ldb = j.ldc = ld;
  if (k.mode == DAG_SYRK) {
    T* Am = static_cast<T*>(t.A);
    j.A = Am + (int64_t)k.r * TM * ld + (int64_t)k.p0 * HBO_TILE;
    j.B = Am + (int64_t)k.c * TM * ld + (int64_t)k.p0 * HBO_TILE;
    j.C = Am + (int64_t)k.r * TM * ld + (int64_t)k.c * TM;
    j.ksteps = k.kt * HBO_TILE / BKE;
    j.alpha = (T)-1; j.beta = 1;
    return;
  }
  // tree node: s = p0 blocks, group = kt; tile (it, jt) = (r, c) in TM units
  const int s = k.p0, grp = k.kt, it = k.r, jt = k.c;
  const int64_t o = (int64_t)grp * 2 * s * HBO_TILE;
  const int64_t R = o + (int64_t)s * HBO_TILE + (int64_t)it * TM;
  const int64_t Cc = o + (int64_t)jt * TM;
  const T* L = static_cast<const T*>(t.A);
  T* W = static_cast<T*>(t.W);
  T* S = static_cast<T*>(t.S);
  if (k.mode == DAG_TRTRI_A) {
    j.A = L + R * ld + Cc;
    j.B = W + Cc * ld + Cc;
    j.C = S + R * ld + Cc;
    j.ksteps = (s * HBO_TILE - jt * TM) / BKE;
    j.alpha = (T)1;
  } else {
    const int64_t o2 = o + (int64_t)s * HBO_TILE;
    j.A = W + R * ld + o2;
    j.B = S + o2 * ld + Cc;
    j.C = W + R * ld + Cc;
    j.ksteps = (it + 1) * TM / BKE;
    j.alpha = (T)-1;
  }
  j.beta = 0;
  (void)U;
}

__device__ __forceinline__ bool dag_ready(const DagTask& k, const int* cm) {
  for (int d = 0; d < k.ndep; ++d)
    if (dag_load(cm + k.dep_idx[d]) < k.dep_thr[d]) return false;
  return true;
}

// Scheduler: run by ALL lanes of wave 0 in lockstep (uniform loads, lane 0 alone issues the atomics).  A single-lane
// scheduler does not survive the compiler: the loop nest is structurised so that lane 0 leaves the task loop through an
// exec-masked exit while lanes 1..63 of its wave go on to the next barrier and wait there for the mailbox lane 0 can no
// longer write.  Returns the index of a task whose dependencies are met (acquired), -1 when every queue is exhausted,
// -2 on abort.  `held[q]` (LDS) = a task drawn from queue q that was not ready yet.
__device__ __forceinline__ int dag_next_task(const DagDev& d, int* held, unsigned long long t_start, int lane) {
  int* const ctr = d.ctr;
  unsigned idle = 0;
  for (;;) {
    if (dag_load(ctr + DAG_ABORT)) return -2;
    bool exhausted = true;
    for (int q = 0; q < DAG_NQ; ++q) {
      int ti = held[q];
      const bool fresh = ti < 0;
      if (fresh) {
        // first segment of the queue that still has tasks to hand out
        int sg = dag_load(ctr + DAG_HINT + q);
        if (sg < d.q_first[q]) sg = d.q_first[q];
        const int sg_end = d.q_first[q + 1];
        int cur = 0;
        while (sg < sg_end && (cur = dag_load(ctr + d.off_cursor + sg)) >= d.segs[sg].len) ++sg;
        if (sg >= sg_end) continue;
        exhausted = false;
        if (lane == 0) __hip_atomic_fetch_max(ctr + DAG_HINT + q, sg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // the segment is open when its next task is ready
        const DagTask& head = d.tasks[d.segs[sg].start + cur];
        if (!dag_ready(head, ctr + d.off_mat + head.mat * d.stride)) continue;
        int got = 0;
        if (lane == 0) got = __hip_atomic_fetch_add(ctr + d.off_cursor + sg, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        got = __builtin_amdgcn_readfirstlane(got);
        if (got >= d.segs[sg].len) { --q; continue; }   // the segment ran out under us: look again
        ti = d.segs[sg].start + got;
#ifdef HBO_DAG_DEBUG
        if (q == 0 && lane == 0) { const DagTask& kk = d.tasks[ti]; dag_stamp_min(d.stamps, kk.p0 + kk.kt - 1, 2); }
#endif
      }
      exhausted = false;
      const DagTask& k = d.tasks[ti];
      const int* cm = ctr + d.off_mat + k.mat * d.stride;
      bool ok = dag_ready(k, cm);
      if (!ok && q == 0 && fresh && d.spin_us > 0) {
        // chain-critical: its inputs are a panel kernel or a neighbouring tile away
        const unsigned long long t0 = wall_clock64();
        while (!ok && (long long)(wall_clock64() - t0) < (long long)d.spin_us * 100 && !dag_load(ctr + DAG_ABORT)) {
          __builtin_amdgcn_s_sleep(2);
          ok = dag_ready(k, cm);
        }
      }
      if (ok) {
        held[q] = -1;
        if (!(d.dbg_flags & 1)) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        return ti;
      }
      held[q] = ti;
    }
    if (exhausted) return -1;
    // nothing to do right now: back off (every idle workgroup re-reads the same few words; hundreds of them polling flat out
    // slow the memory operations of the panel kernels they are waiting for)
    for (int i = 0; i < d.idle_sleep; ++i) __builtin_amdgcn_s_sleep(32);
    if ((++idle & 255) == 0 && (long long)(wall_clock64() - t_start) > d.timeout_ticks) {
      if (lane == 0) __hip_atomic_store(ctr + DAG_ABORT, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return -2;
    }
  }
}

#ifdef HBO_DAG_INLINE
template <typename T, bool BKC, int TM>
__device__ __forceinline__ void dag_run_tile(const TileJob<T>& job) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_tile[];
  gemm_tile<T, true, BKC, TM>(job, smem_tile);
}
#else
template <typename T, bool BKC, int TM>
__device__ __noinline__ void dag_run_tile(TileJob<T> job) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_tile[];
  gemm_tile<T, true, BKC, TM>(job, smem_tile);
}
#endif

template <typename T>
__global__ __launch_bounds__(256, 2) void dag_worker_kernel(DagDev d) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  int* const mailbox = reinterpret_cast<int*>(smem + GEMM_LDS_BYTES);
  int* const ctr = d.ctr;
  const int tid = threadIdx.x;
  int* const held = mailbox + 4;   // scheduler state lives in LDS: lane 0's registers belong to the tile
  if (tid == 0) {
    for (int q = 0; q < DAG_NQ; ++q) held[q] = -1;
    *reinterpret_cast<unsigned long long*>(mailbox + 2) = wall_clock64();
    // the first `reserve` CUs of every shader engine to register belong to the panel kernels
    int go = 1;
    if (d.reserve > 0) {
      const int tok = cu_token();
      const int before = __hip_atomic_fetch_add(ctr + d.off_reg + tok, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      int dec;
      if (before == 0) {
        __hip_atomic_fetch_add(ctr + DAG_NSEEN, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int ord = __hip_atomic_fetch_add(ctr + DAG_SE_SEEN + (((tok - 1) >> 4) & 255), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        dec = ord < d.reserve ? 2 : 1;
        __hip_atomic_store(ctr + d.off_dec + tok, dec, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        dec = 0;
        for (int spin = 0; spin < (1 << 20) && (dec = dag_load(ctr + d.off_dec + tok)) == 0; ++spin) __builtin_amdgcn_s_sleep(1);
      }
      go = dec != 2;
    }
    mailbox[0] = go;
  }
  __syncthreads();
  if (!mailbox[0]) return;
  __syncthreads();
#ifdef HBO_DAG_DEBUG
  unsigned long long tk_sched = 0, tk_tile = 0, tk_pub = 0, n_mine = 0, tk0 = wall_clock64();
#endif
  for (;;) {
    if (tid < 64) {   // wave 0, all of it (see dag_next_task)
      const int nxt = dag_next_task(d, held, *reinterpret_cast<unsigned long long*>(mailbox + 2), tid);
      mailbox[0] = nxt;
    }
    __syncthreads();
#ifdef HBO_DAG_DEBUG
    { const unsigned long long now = wall_clock64(); tk_sched += now - tk0; tk0 = now; }
#endif
    const int ti = __builtin_amdgcn_readfirstlane(mailbox[0]);
    __syncthreads();
    if (ti < 0) break;
    const DagTask k = d.tasks[ti];
    const TaskDesc& t = d.mats[k.mat];
    TileJob<T> job;
    const bool urgent = k.mode == DAG_SYRK && k.c < (k.p0 / d.q_group + 2) * d.q_group * (k.tm64 ? 2 : 1);
    if (urgent && tid == 0) dag_stamp_min(d.stamps, k.p0 + k.kt - 1, 3);
    if (k.tm64) {
      decode_dag_task<T, 64>(k, t, job);
      if (k.mode == DAG_SYRK) dag_run_tile<T, true, 64>(job);
      else dag_run_tile<T, false, 64>(job);
    } else {
      decode_dag_task<T, 128>(k, t, job);
      if (k.mode == DAG_SYRK) dag_run_tile<T, true, 128>(job);
      else dag_run_tile<T, false, 128>(job);
    }
#ifdef HBO_DAG_DEBUG
    { const unsigned long long now = wall_clock64(); tk_tile += now - tk0; tk0 = now; ++n_mine; }
#endif
    if (d.dbg_flags & 2) { __syncthreads(); if (tid == 0) dag_bump(ctr + d.off_mat + k.mat * d.stride + k.done_idx, k.done_inc); }
    else dag_wg_publish(ctr + d.off_mat + k.mat * d.stride, k.done_idx, k.done_inc);
    if (urgent && tid == 0) dag_stamp_max(d.stamps, k.p0 + k.kt - 1, 4);
#ifdef HBO_DAG_DEBUG
    { const unsigned long long now = wall_clock64(); tk_pub += now - tk0; tk0 = now; }
#endif
  }
#ifdef HBO_DAG_DEBUG
  if (tid == 0 && d.stamps) {
    unsigned long long* st = d.stamps + 120 * DAG_STAMPS_PER_PANEL;
    __hip_atomic_fetch_add(st + 0, tk_sched, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(st + 1, tk_tile, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(st + 2, tk_pub, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(st + 3, n_mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(st + 4, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
#endif
}

// ---- host: the task lists -------------------------------------------------------------------------------------------

struct DagHost {
  std::vector<DagTask> tasks;
  std::vector<DagSeg> segs;
  int q_first[DAG_NQ + 1];
  DagLayout lay;
  int nmat = 0, max_nblk = 0, q = 0;
  int n_ctr = 0, off_cursor = 0, off_reg = 0, off_dec = 0, off_mat = 0;
  std::vector<int> need;          // VER threshold of column p (same for every matrix)
  std::vector<int> trtri_done;    // per matrix: 1 when the whole inverse is inside the task lists
  double flops_syrk = 0, flops_trtri = 0;
  // device copies
  DagTask* d_tasks = nullptr; DagSeg* d_segs = nullptr; int* d_ctr = nullptr;
};

struct DagKey {
  std::vector<int> nblk; int q, near64, trtri, small_nblk, chain_tasks;
  bool operator<(const DagKey& o) const {
    if (q != o.q) return q < o.q;
    if (near64 != o.near64) return near64 < o.near64;
    if (trtri != o.trtri) return trtri < o.trtri;
    if (small_nblk != o.small_nblk) return small_nblk < o.small_nblk;
    if (chain_tasks != o.chain_tasks) return chain_tasks < o.chain_tasks;
    return nblk < o.nblk;
  }
};

// updates a tile of column c has received when its panel step comes: one per group that starts before c
inline int updates_of_column(int c, int q) { return (c + q - 1) / q; }

struct DagBuilder {
  DagHost& h; int q, near64, small_nblk;
  std::vector<DagTask> cur;
  void add_syrk(int mat, int nblk, int r, int c, int p0, int kt, bool quads) {
    // tile (r, c) in 128-units, update number = group index of p0
    const DagLayout& L = h.lay;
    const int upd = p0 / q, plast = p0 + kt - 1;
    DagTask k; memset(&k, 0, sizeof k);
    k.mode = DAG_SYRK; k.mat = (short)mat; k.p0 = (short)p0; k.kt = (short)kt;
    int nd = 0;
    k.dep_idx[nd] = L.row(plast, r); k.dep_thr[nd++] = 2;
    if (c != r) { k.dep_idx[nd] = L.row(plast, c); k.dep_thr[nd++] = 2; }
    if (upd > 0) { k.dep_idx[nd] = L.ver(r, c); k.dep_thr[nd++] = DAG_VER_UNIT * upd; }
    k.ndep = (unsigned char)nd;
    k.done_idx = L.ver(r, c);
    (void)nblk;
    h.flops_syrk += 2.0 * 128 * 128 * 128 * kt;
    if (quads) {
      k.tm64 = 1; k.done_inc = 1;
      for (int qr = 0; qr < 2; ++qr) for (int qc = 0; qc < 2; ++qc) { k.r = (short)(2 * r + qr); k.c = (short)(2 * c + qc); cur.push_back(k); }
    } else {
      k.tm64 = 0; k.done_inc = DAG_VER_UNIT; k.r = (short)r; k.c = (short)c;
      cur.push_back(k);
    }
  }
  void close_segment() {
    if (cur.empty()) return;
    DagSeg s; s.start = (int)h.tasks.size(); s.len = (int)cur.size();
    h.tasks.insert(h.tasks.end(), cur.begin(), cur.end());
    h.segs.push_back(s);
    cur.clear();
  }
};

// inverse tree of a matrix of nblk blocks: node (li, g): s = 2^li, blocks [2sg, 2sg + 2s); exists iff 2sg + s < nblk
struct TreeDep { int idx, thr; };
inline TreeDep subtree_done(const DagLayout& L, int nblk, int li, int g, int U) {
  // li == -1: the single block g
  for (;;) {
    if (li < 0) return {L.diag(g), 2};
    const int s = 1 << li, o = 2 * s * g;
    if (o + s < nblk) { const int v = std::min(s, nblk - (o + s)); return {L.nb(li, g), v * U * s * U}; }
    --li; g *= 2;
  }
}

// chain_tasks: the chain-critical updates (queue 0) are tasks too; false: the panel stream launches them itself (run_potrf_dag)
void build_dag(DagHost& h, const std::vector<int>& nblk, int q, int near64, int trtri_cut, int small_nblk, bool chain_tasks) {
  const int T = (int)nblk.size();
  int M = 0; for (int n : nblk) M = std::max(M, n);
  h.nmat = T; h.max_nblk = M; h.q = q;
  h.lay = dag_layout(M);
  h.need.resize(M);
  for (int p = 0; p < M; ++p) h.need[p] = chain_tasks ? DAG_VER_UNIT * updates_of_column(p, q) : 0;
  h.trtri_done.assign(T, 0);
  DagBuilder b{h, q, near64, small_nblk};

  // queue 0: after the panel solve of panel p -- the next column of the same group (left-looking inside the group), or, when
  // p closes its group, the block columns of the next group; most urgent tile (next diagonal block) first
  h.q_first[0] = 0;
  for (int p = 0; p < M && chain_tasks; ++p) {
    const int g0 = (p / q) * q;
    const bool closes = (p + 1) % q == 0;
    // rank-major over the matrices so that every matrix's urgent tiles come first
    std::vector<std::vector<std::array<int, 5>>> per(T);
    for (int t = 0; t < T; ++t) {
      const int n = nblk[t];
      if (p + 1 >= n) continue;
      if (!closes) { const int c = p + 1; for (int r = c; r <= n; ++r) per[t].push_back({r, c, g0, p + 1 - g0, 0}); }
      else for (int c = p + 1; c < std::min(p + 1 + q, n); ++c) for (int r = c; r <= n; ++r) per[t].push_back({r, c, g0, q, 0});
    }
    size_t longest = 0; for (auto& v : per) longest = std::max(longest, v.size());
    for (size_t i = 0; i < longest; ++i)
      for (int t = 0; t < T; ++t) if (i < per[t].size()) {
        const auto& e = per[t][i];
        // 64x64 quadrants for the tiles the next panel kernels wait for first (or for all of them: near64 = 2)
        const bool quads = near64 >= 2 || (near64 == 1 && (e[0] - e[1] <= 1 || nblk[t] <= small_nblk));
        b.add_syrk(t, nblk[t], e[0], e[1], e[2], e[3], quads);
      }
    b.close_segment();
  }
  h.q_first[1] = (int)h.segs.size();
  // queue 1: the bulk of every group's trailing update (block columns beyond the next group), column-major
  for (int g0 = 0; g0 + q <= M; g0 += q) {
    const int g2 = g0 + 2 * q;
    for (int t = 0; t < T; ++t) {
      const int n = nblk[t];
      if (g0 + q > n) continue;
      for (int c = g2; c < n; ++c) for (int r = c; r <= n; ++r) b.add_syrk(t, n, r, c, g0, q, n <= small_nblk && near64 >= 1);
    }
    b.close_segment();
  }
  h.q_first[2] = (int)h.segs.size();
  // queue 2: the block-recursive inverse in the order its pieces become computable (cfin = block columns of L that are final)
  if (trtri_cut > 0) {
    const DagLayout& L = h.lay;
    for (int cfin = 1; cfin <= M; ++cfin) {
      for (int li = 0, s = 1; s < M; ++li, s *= 2) {
        for (int phase = 0; phase < 2; ++phase) {   // A, then B
          for (int t = 0; t < T; ++t) {
            const int n = nblk[t];
            if ((int64_t)cfin * 64 > (int64_t)trtri_cut * n) continue;   // beyond the cut: left to the launches after the factorisation
            const bool small = n <= small_nblk;
            const int U = small ? 2 : 1;
            for (int g = 0; 2 * s * g + s < n; ++g) {
              const int o = 2 * s * g, v = std::min(s, n - (o + s));
              const int avail = phase == 0 ? o + s : std::min(o + 2 * s, n);
              if (avail != cfin) continue;
              DagTask k; memset(&k, 0, sizeof k);
              k.mat = (short)t; k.p0 = (short)s; k.kt = (short)g; k.tm64 = small ? 1 : 0; k.done_inc = 1;
              if (phase == 0) {
                k.mode = DAG_TRTRI_A; k.done_idx = L.na(li, g);
                const TreeDep left = subtree_done(L, n, li - 1, 2 * g, U);
                // long K first: jt ascending (K = s*128 - jt*TM), rows inner
                for (int jt = 0; jt < s * U; ++jt) for (int it = 0; it < v * U; ++it) {
                  k.r = (short)it; k.c = (short)jt;
                  k.ndep = 2;
                  k.dep_idx[0] = L.row(o + s - 1, o + s + it / U); k.dep_thr[0] = 2;
                  k.dep_idx[1] = left.idx; k.dep_thr[1] = left.thr;
                  b.cur.push_back(k);
                  h.flops_trtri += 2.0 * (128 / U) * (128 / U) * (s * 128 - jt * (128 / U));
                }
              } else {
                k.mode = DAG_TRTRI_B; k.done_idx = L.nb(li, g);
                const TreeDep right = subtree_done(L, n, li - 1, 2 * g + 1, U);
                for (int it = v * U - 1; it >= 0; --it) for (int jt = 0; jt < s * U; ++jt) {
                  k.r = (short)it; k.c = (short)jt;
                  k.ndep = 2;
                  k.dep_idx[0] = L.na(li, g); k.dep_thr[0] = v * U * s * U;
                  k.dep_idx[1] = right.idx; k.dep_thr[1] = right.thr;
                  b.cur.push_back(k);
                  h.flops_trtri += 2.0 * (128 / U) * (128 / U) * ((it + 1) * (128 / U));
                }
              }
            }
          }
          b.close_segment();
        }
      }
    }
    for (int t = 0; t < T; ++t) h.trtri_done[t] = trtri_cut >= 64;
  }
  h.q_first[3] = (int)h.segs.size();
  // counters
  h.off_cursor = DAG_FIXED;
  h.off_reg = h.off_cursor + (int)h.segs.size();
  h.off_dec = h.off_reg + DAG_CU_TAB;
  h.off_mat = h.off_dec + DAG_CU_TAB;
  h.n_ctr = h.off_mat + T * h.lay.stride;
}

std::map<DagKey, DagHost>& dag_cache(hbo_ctx* c) {
  static std::map<hbo_ctx*, std::map<DagKey, DagHost>> all;
  return all[c];
}

}  // namespace

// progress of the inverse tree once the task lists up to `cut` (64ths of the block count) have run: what trtri_advance has
// to skip afterwards
static void mark_trtri_progress(TrtriProgress& pg, int nblk, int cut) {
  const int cfin = (int)((int64_t)cut * nblk / 64);
  pg.diag = std::max(pg.diag, cfin);
  int li = 0;
  for (int s = 1; s < nblk && li < 12; s *= 2, ++li) {
    const int ngrp = (nblk - s + 2 * s - 1) / (2 * s);
    int na = cfin >= s ? (cfin - s) / (2 * s) + 1 : 0;
    int nb = cfin >= nblk ? ngrp : cfin / (2 * s);
    pg.a[li] = std::max(pg.a[li], std::min(na, ngrp));
    pg.b[li] = std::max(pg.b[li], std::min(nb, ngrp));
  }
}

bool run_potrf_dag(hbo_ctx* c, int dtype, const TaskDesc* d_tasks, int ntasks, int max_nblk, const int* h_nblk, int* d_info,
                   TrtriProgress* early) {
  std::vector<int> nblk(ntasks);
  for (int t = 0; t < ntasks; ++t) nblk[t] = h_nblk ? h_nblk[t] : max_nblk;
  const int q = c->opt_group > 0 ? c->opt_group : 3;
  const int cut = early ? c->opt_dag_trtri : 0;
  const bool chain_tasks = c->opt_dag == 2;
  DagKey key{nblk, q, c->opt_dag_near64, cut, c->opt_small_nblk, chain_tasks ? 1 : 0};
  auto& cache = dag_cache(c);
  auto it = cache.find(key);
  if (it == cache.end()) {
    if (cache.size() >= 16) {   // shapes come and go (sub-sampled training batches): keep the table small
      for (auto& kv : cache) { hipFree(kv.second.d_tasks); hipFree(kv.second.d_segs); hipFree(kv.second.d_ctr); }
      cache.clear();
    }
    DagHost h;
    build_dag(h, nblk, q, c->opt_dag_near64, cut, c->opt_small_nblk, chain_tasks);
    if (hbo_malloc(c, (void**)&h.d_tasks, sizeof(DagTask) * std::max<size_t>(h.tasks.size(), 1)) != hipSuccess) return false;
    if (hbo_malloc(c, (void**)&h.d_segs, sizeof(DagSeg) * std::max<size_t>(h.segs.size(), 1)) != hipSuccess) return false;
    if (hbo_malloc(c, (void**)&h.d_ctr, sizeof(int) * h.n_ctr) != hipSuccess) return false;
    hipMemcpy(h.d_tasks, h.tasks.data(), sizeof(DagTask) * h.tasks.size(), hipMemcpyHostToDevice);
    hipMemcpy(h.d_segs, h.segs.data(), sizeof(DagSeg) * h.segs.size(), hipMemcpyHostToDevice);
    it = cache.emplace(key, std::move(h)).first;
  }
  DagHost& h = it->second;
  hipStream_t sm = c->stream, sp = c->stream2, sd = c->stream4;
  hipMemsetAsync(h.d_ctr, 0, sizeof(int) * h.n_ctr, sm);
  size_t evi = 8;
  { hipEvent_t e = pool_event(c, evi++); hipEventRecord(e, sm); hipStreamWaitEvent(sp, e, 0); hipStreamWaitEvent(sd, e, 0); }

  DagDev d; memset(&d, 0, sizeof d);
  d.mats = d_tasks; d.tasks = h.d_tasks; d.segs = h.d_segs; d.ctr = h.d_ctr;
  for (int i = 0; i <= DAG_NQ; ++i) d.q_first[i] = h.q_first[i];
  d.off_cursor = h.off_cursor; d.off_reg = h.off_reg; d.off_dec = h.off_dec; d.off_mat = h.off_mat; d.stride = h.lay.stride;
  d.reserve = c->opt_dag_reserve;
  d.spin_us = c->opt_dag_spin_us;
  d.dbg_flags = c->opt_dag_dbg;
  d.idle_sleep = c->opt_dag_idle_sleep;
  d.q_group = q;
  d.stamps = nullptr;
#ifdef HBO_DAG_DEBUG
  {
    static unsigned long long* d_st = nullptr;
    if (!d_st) hbo_malloc(c, (void**)&d_st, sizeof(unsigned long long) * 128 * DAG_STAMPS_PER_PANEL);
    std::vector<unsigned long long> init(128 * DAG_STAMPS_PER_PANEL);
    for (int i = 0; i < 128 * DAG_STAMPS_PER_PANEL; ++i) { const int k = i % DAG_STAMPS_PER_PANEL; init[i] = (k == 1 || k == 4 || k == 7 || k == 9 || i >= 120 * DAG_STAMPS_PER_PANEL) ? 0ull : ~0ull; }
    hipMemcpyAsync(d_st, init.data(), sizeof(unsigned long long) * init.size(), hipMemcpyHostToDevice, sm);
    hipStreamSynchronize(sm);
    d.stamps = d_st; g_dag_stamps = d_st;
  }
#endif
  d.timeout_ticks = (long long)c->opt_dag_timeout_ms * 100000;
  {
    ProfScope ps(c, "dag_worker", 1, sm);
    static bool attr = false;
    if (!attr) {
      hipFuncSetAttribute(reinterpret_cast<const void*>(&dag_worker_kernel<double>), hipFuncAttributeMaxDynamicSharedMemorySize, DAG_LDS_BYTES);
      hipFuncSetAttribute(reinterpret_cast<const void*>(&dag_worker_kernel<float>), hipFuncAttributeMaxDynamicSharedMemorySize, DAG_LDS_BYTES);
      attr = true;
    }
    if (dtype == HBO_F64) hipLaunchKernelGGL((dag_worker_kernel<double>), dim3(2 * c->n_cus), dim3(256), DAG_LDS_BYTES, sm, d);
    else hipLaunchKernelGGL((dag_worker_kernel<float>), dim3(2 * c->n_cus), dim3(256), DAG_LDS_BYTES, sm, d);
  }
  c->dag_flops = h.flops_syrk + h.flops_trtri;

  ChainSync cs; memset(&cs, 0, sizeof cs);
  cs.stamps = d.stamps;
  cs.ctr = h.d_ctr; cs.off_mat = h.off_mat; cs.stride = h.lay.stride; cs.lay = h.lay; cs.timeout_ticks = d.timeout_ticks;
  const int dstep = 4;   // diagonal-block inverses: every fourth panel, on the side stream
  int diag_done = 0;
  auto after_panel = [&](int p) {
    if (cut > 0 && ((p + 1) % dstep == 0 || p + 1 == max_nblk)) {
      // (needs only potf2 of its blocks; the event sits behind the panel solve, which is a few microseconds later)
      hipEvent_t e = pool_event(c, evi++); hipEventRecord(e, sp); hipStreamWaitEvent(sd, e, 0);
      ProfScope ps(c, "trtri_diag", 2, sd);
      launch_trtri_diag(dtype, d_tasks, ntasks, diag_done, p + 1, sd, &cs);
      diag_done = p + 1;
    }
  };
  if (chain_tasks) {
    for (int p = 0; p < max_nblk; ++p) {
      cs.need = h.need[p];
      { ProfScope ps(c, "potf2", 2, sp); launch_potf2(dtype, d_tasks, ntasks, p, d_info, sp, nullptr, &cs); }
      { ProfScope ps(c, "trsm", 2, sp); launch_trsm(dtype, d_tasks, ntasks, p, max_nblk, sp, nullptr, &cs); }
      after_panel(p);
    }
  } else {
    // The panel stream keeps its own launches -- left-looking column updates inside a group, then the next group's block
    // columns (F1) -- on the CUs the tile workgroups left; only the bulk of every trailing update and the inverse are tasks.
    // What crosses between the two: the panel solve counts its row blocks done (tasks of that group wait for them), and an
    // F1 tile waits until the earlier groups' bulk updates have reached it.
    cs.need = 0;
    for (int g0 = 0; g0 < max_nblk; g0 += q) {
      const int g1 = std::min(g0 + q, max_nblk), g2 = std::min(g1 + q, max_nblk);
      for (int p = g0; p < g1; ++p) {
        if (p > g0) {
          ProfScope ps(c, "syrk_col", 2, sp);
          GemmArgs a = {}; a.tasks = d_tasks; a.mode = GEMM_SYRK; a.p0 = g0; a.kt = p - g0; a.c_lo = p; a.c_hi = p + 1; a.aug = 1; a.small_tiles = 1;
          launch_gemm(dtype, a, dim3(max_nblk + 1 - p, 1, ntasks), sp);
        }
        { ProfScope ps(c, "potf2", 2, sp); launch_potf2(dtype, d_tasks, ntasks, p, d_info, sp, nullptr, nullptr); }
        { ProfScope ps(c, "trsm", 2, sp); launch_trsm(dtype, d_tasks, ntasks, p, max_nblk, sp, nullptr, &cs); }
        after_panel(p);
      }
      if (g1 < max_nblk) {
        ProfScope ps(c, "syrk_trailing", 1, sp);
        GemmArgs a = {}; a.tasks = d_tasks; a.mode = GEMM_SYRK; a.p0 = g0; a.kt = g1 - g0; a.aug = 1;
        a.c_lo = g1; a.c_hi = g2;
        a.small_tiles = (int64_t)(max_nblk + 1 - a.c_lo) * (a.c_hi - a.c_lo) * ntasks < c->opt_dag_f1_small;
        a.dag_ctr = h.d_ctr; a.dag_off = h.off_mat + h.lay.off_ver; a.dag_stride = h.lay.stride; a.dag_M = h.lay.M;
        a.dag_need = DAG_VER_UNIT * (g0 / q); a.dag_timeout = d.timeout_ticks;
        launch_gemm(dtype, a, dim3(max_nblk + 1 - a.c_lo, a.c_hi - a.c_lo, ntasks), sp);
      }
    }
    if (c->opt_dag_join) {
      // the panel chain is done: its CUs join the tile workgroups for what is left of the bulk updates and the inverse
      // (behind the last diagonal-block inverse: that kernel needs one of these CUs, and the tasks wait for its counter)
      { hipEvent_t e = pool_event(c, evi++); hipEventRecord(e, sd); hipStreamWaitEvent(sp, e, 0); }
      DagDev d2 = d; d2.reserve = 0;
      ProfScope ps(c, "dag_join", 2, sp);
      if (dtype == HBO_F64) hipLaunchKernelGGL((dag_worker_kernel<double>), dim3(2 * 8 * 4 * c->opt_dag_reserve), dim3(256), DAG_LDS_BYTES, sp, d2);
      else hipLaunchKernelGGL((dag_worker_kernel<float>), dim3(2 * 8 * 4 * c->opt_dag_reserve), dim3(256), DAG_LDS_BYTES, sp, d2);
    }
  }
  { hipEvent_t e = pool_event(c, evi++); hipEventRecord(e, sp); hipStreamWaitEvent(sm, e, 0); }
  { hipEvent_t e = pool_event(c, evi++); hipEventRecord(e, sd); hipStreamWaitEvent(sm, e, 0); }
  c->dag_ctr_last = h.d_ctr;
#ifdef HBO_DAG_DEBUG
  if (getenv("HBO_DAG_WATCH")) {
    // bring-up aid: HBO_DAG_WATCH=<ms> prints the counters of a run that does not come back (a copy on its own stream passes
    // the spinning kernels): abort word, segment cursors against their lengths, the first counters of matrix 0
    DagHost* hp = &h;
    const int dev = c->device;
    std::thread([hp, dev]() {
      hipSetDevice(dev);
      std::this_thread::sleep_for(std::chrono::milliseconds(atoi(getenv("HBO_DAG_WATCH"))));
      hipStream_t s2; hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
      std::vector<int> v(hp->n_ctr);
      hipMemcpyAsync(v.data(), hp->d_ctr, sizeof(int) * hp->n_ctr, hipMemcpyDeviceToHost, s2);
      hipStreamSynchronize(s2);
      fprintf(stderr, "[dag watch] abort %d CUs seen %d hints %d %d %d  segs %zu tasks %zu\n", v[DAG_ABORT], v[DAG_NSEEN], v[DAG_HINT], v[DAG_HINT + 1], v[DAG_HINT + 2], hp->segs.size(), hp->tasks.size());
      for (int q = 0; q < DAG_NQ; ++q) {
        fprintf(stderr, "[dag watch] queue %d (drawn/len):", q);
        for (int sg = hp->q_first[q]; sg < hp->q_first[q + 1] && sg < hp->q_first[q] + 24; ++sg) fprintf(stderr, " %d/%d", v[hp->off_cursor + sg], hp->segs[sg].len);
        fprintf(stderr, "\n");
      }
      const DagLayout& L = hp->lay;
      const int* m0 = v.data() + hp->off_mat;
      fprintf(stderr, "[dag watch] ver(p, p):"); for (int p = 0; p < std::min(L.M, 16); ++p) fprintf(stderr, " %d", m0[L.ver(p, p)]); fprintf(stderr, "\n");
      fprintf(stderr, "[dag watch] row(p, p+1):"); for (int p = 0; p + 1 < std::min(L.M, 16); ++p) fprintf(stderr, " %d", m0[L.row(p, p + 1)]); fprintf(stderr, "\n");
      fprintf(stderr, "[dag watch] diag(p):"); for (int p = 0; p < std::min(L.M, 16); ++p) fprintf(stderr, " %d", m0[L.diag(p)]); fprintf(stderr, "\n");
      fflush(stderr);
    }).detach();
  }
#endif
  if (early && cut > 0) {
    // what the task lists covered is done when the worker kernel ends; with ragged batches the cut is per matrix, and the
    // launch path works on the largest one: only a complete inverse is marked
    if (cut >= 64) mark_trtri_progress(*early, max_nblk, 64);
    else if (ntasks == 1) mark_trtri_progress(*early, max_nblk, cut);
  }
  return true;
}

#ifdef HBO_DAG_DEBUG
extern "C" void hbo_dbg_dag_stamps(unsigned long long* host) { if (g_dag_stamps) hipMemcpy(host, g_dag_stamps, sizeof(unsigned long long) * 128 * DAG_STAMPS_PER_PANEL, hipMemcpyDeviceToHost); }
#endif
bool dag_aborted(hbo_ctx* c) {
  if (!c->dag_ctr_last) return false;
  int flag = 0;
  hipMemcpy(&flag, c->dag_ctr_last + DAG_ABORT, sizeof(int), hipMemcpyDeviceToHost);
  c->dag_ctr_last = nullptr;
  if (flag) c->dag_broken = 1;
  return flag != 0;
}
