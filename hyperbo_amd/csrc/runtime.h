// Host-side runtime helpers shared by the translation units of libhbo (api.hip: the C ABI; sched.hip: the launch schedules):
// element sizes and padded leading dimensions, grow-only workspaces, the pooled device buffers of datasets and caches, and
// the HIP-event profiling scopes.
#pragma once
#include "ctx.h"

#include <string.h>

#include <string>

static inline size_t esize(int dtype) { return dtype == HBO_F64 ? 8 : 4; }
static inline int round_up(int64_t n, int q) { return (int)(((n + q - 1) / q) * q); }
// leading dimension: padded extent + 128 bytes, so that rows do not sit at a power-of-two stride
// (a 64 KiB row stride maps every row of a k-contiguous tile onto the same L2/HBM channel)
static inline int64_t padded_ld(int64_t extent, int dtype) { return extent + 128 / (int64_t)esize(dtype); }

// Every device allocation of the library: out of memory with buffers parked in the pool (dev_free below) -> release them and
// retry once.  (A training loop with changing batch shapes parks gigabytes; a posterior workspace must not fail beside them.)
static inline void pool_release_all(hbo_ctx* c);
static inline hipError_t hbo_malloc(hbo_ctx* c, void** out, size_t bytes) {
  hipError_t e = hipMalloc(out, bytes ? bytes : 16);
  if (e == hipErrorOutOfMemory && c && c->pool_bytes) {
    (void)hipGetLastError();
    pool_release_all(c);
    e = hipMalloc(out, bytes ? bytes : 16);
  }
  return e;
}

// grow-only scratch buffer for `slot`; nullptr on allocation failure (ctx->err is set)
#define HBO_H2_LEVELS 16
#define HBO_H2_WORDS (3 * HBO_H2_LEVELS + 4)   // per level of the inverse: max |W11|, |S21|, |W22|; the last word: max |W| for K^-1 = W^T W
enum WsSlot { WS_XQ = 1, WS_MU0, WS_KD, WS_MU, WS_VAR, WS_ACQ, WS_K, WS_COLSQ, WS_V, WS_KQQ, WS_COV,
              WS_AP_KX, WS_AP_L, WS_AP_W, WS_AP_MU, WS_AP_KD, WS_AP_Y, WS_VPART, WS_GATHER, WS_COUNTERS, WS_MUPART,
              WS_AG_K, WS_AG_L, WS_AG_B, WS_AG_GF, WS_AG_DMU, WS_AG_GX, WS_AG_T0, WS_AG_T1, WS_AG_DW, WS_SHARD_RED, WS_SHARD_MAP, WS_SYRK3_A, WS_SYRK3_B, WS_TRTRI3_X, WS_TRTRI3_Y, WS_LAUUM3, WS_SMP_MODELS, WS_SMP_DESC, WS_SMP_INFO, WS_SMP_ACQ, WS_SMP_MLP, WS_SMP_XQ, WS_EXTRA_Z, WS_SMALL_W, WS_H2_AUG, WS_H2_SCALES, WS_FQ0 /* + layer */,
              WS_FQ_LAST = WS_FQ0 + HBO_MAX_MLP_LAYERS - 1, WS_K3, WS_SLOT_END };
// every slot is distinct by construction (auto-numbered; the per-layer range WS_FQ0.. is closed by WS_FQ_LAST before WS_K3), and the
// posterior's per-lane offset (cache.hip: 4096 * lane) must clear the whole range
static_assert(WS_FQ0 + HBO_MAX_MLP_LAYERS <= WS_K3 && WS_SLOT_END < 4096, "workspace slots overlap a lane offset");
// hbo_tune "poison" (tests): the numeric scratch buffers -- fetched once per call and fully rewritten before they are read -- come back
// filled with NaN, so that a product that skips tiles cannot hide behind what the previous call left there.  (Synchronous memset on the
// null stream: ordered against every stream of the context.)  Tables, counters and reduction buffers that a call fetches again while
// they hold live data are not in the list.
static inline bool ws_poisonable(int slot) {
  switch (slot % 4096) {
    case WS_MU0: case WS_KD: case WS_MU: case WS_VAR: case WS_ACQ: case WS_K: case WS_COLSQ: case WS_V: case WS_KQQ: case WS_COV:
    case WS_VPART: case WS_MUPART: case WS_K3: case WS_SYRK3_A: case WS_SYRK3_B: case WS_TRTRI3_X: case WS_TRTRI3_Y: case WS_LAUUM3:
    case WS_AG_K: case WS_AG_L: case WS_AG_B: case WS_EXTRA_Z:
      return true;
    default: return false;
  }
}
static inline void* ws_get(hbo_ctx* c, int slot, size_t bytes) {
  auto& e = c->ws[slot];
  if (e.second < bytes || !e.first) {
    if (e.first) hipFree(e.first);
    e.first = nullptr; e.second = 0;
    hipError_t err = hbo_malloc(c, &e.first, bytes);
    if (err != hipSuccess) { c->err = std::string("hipMalloc failed: ") + hipGetErrorString(err); e.first = nullptr; return nullptr; }
    e.second = bytes;
  }
  if (c->opt_poison && bytes && ws_poisonable(slot)) hipMemset(e.first, 0xFF, bytes);
  return e.first;
}

// ---- pooled device buffers of datasets and caches (see hbo_ctx::pool_free) ---------------------
static inline void pool_release_all(hbo_ctx* c) {
  for (auto& kv : c->pool_free) for (void* p : kv.second) hipFree(p);
  c->pool_free.clear(); c->pool_bytes = 0;
}
static inline hipError_t dev_alloc(hbo_ctx* c, void** out, size_t bytes, int cls = 0, bool* reused = nullptr) {
  if (reused) *reused = false;
  if (bytes == 0) bytes = 16;
  if (c) {
    auto it = c->pool_free.find({cls, bytes});
    if (it != c->pool_free.end() && !it->second.empty()) {
      *out = it->second.back(); it->second.pop_back();
      c->pool_bytes -= bytes;
      c->pool_live[*out] = {cls, bytes};
      if (reused) *reused = true;
      return hipSuccess;
    }
  }
  hipError_t e = hbo_malloc(c, out, bytes);
  if (e == hipSuccess && c) c->pool_live[*out] = {cls, bytes};
  return e;
}
static inline void dev_free(hbo_ctx* c, void* p) {
  if (!p) return;
  if (c) {
    auto it = c->pool_live.find(p);
    if (it != c->pool_live.end()) {
      const std::pair<int, size_t> key = it->second;
      c->pool_live.erase(it);
      if (c->pool_bytes + key.second <= c->pool_cap) { c->pool_free[key].push_back(p); c->pool_bytes += key.second; return; }
    }
  }
  hipFree(p);
}

// three timing-enabled events of the context (the pool of sched.hip's pool_event is hipEventDisableTiming)
static inline hipEvent_t pool_event_timed(hbo_ctx* c, int i) {
  if (!c->ev_timed[i]) hipEventCreate(&c->ev_timed[i]);
  return c->ev_timed[i];
}

// ---- profiling ---------------------------------------------------------------------------
// Timing scopes: HIP events recorded on the stream the kernels are launched on.  Events come from a pool owned by
// the context (creating and destroying ~80 events per evaluation cost 0.4 ms of host time).  prof_level < 0 is the
// "roofline only" mode of bench.py: just the launches of the dominant kernel (scopes named syrk_bulk) are bracketed.
static inline hipEvent_t prof_event(hbo_ctx* c) {
  if (c->prof_next == c->prof_events.size()) {
    hipEvent_t ev;
    hipEventCreate(&ev);
    c->prof_events.push_back(ev);
  }
  return c->prof_events[c->prof_next++];
}
struct ProfScope {
  hbo_ctx* c; bool on; ProfEntry e; hipStream_t st;
  ProfScope(hbo_ctx* ctx, const char* name, int level, hipStream_t stream = nullptr)
      : c(ctx), on(ctx->prof_level >= level || (ctx->prof_level < 0 && !strcmp(name, "syrk_bulk"))),
        st(stream ? stream : ctx->stream) {
    if (!on) return;
    e.name = name;
    e.e0 = prof_event(c); e.e1 = prof_event(c);
    hipEventRecord(e.e0, st);
  }
  ~ProfScope() {
    if (!on) return;
    hipEventRecord(e.e1, st);
    c->prof_pending.push_back(e);
  }
};
static inline void prof_begin(hbo_ctx* c) {
  c->prof_names.clear(); c->prof_ms.clear(); c->prof_count.clear();
  c->prof_pending.clear();
  c->prof_next = 0;
}
static inline void prof_collect(hbo_ctx* c) {  // stream must be synchronised
  for (auto& p : c->prof_pending) {
    float ms = 0;
    hipEventElapsedTime(&ms, p.e0, p.e1);
    size_t k = 0;
    for (; k < c->prof_names.size(); ++k) if (c->prof_names[k] == p.name) break;
    if (k == c->prof_names.size()) { c->prof_names.push_back(p.name); c->prof_ms.push_back(0); c->prof_count.push_back(0); }
    c->prof_ms[k] += ms; c->prof_count[k] += 1;
  }
  c->prof_pending.clear();
}
