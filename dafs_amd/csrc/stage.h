// dafs_amd/csrc/stage.h -- per-kernel device timings for bench.py's "stages" object (dafs_hip_stage_timing /
// dafs_hip_stage_report).  When a context has switched the recorder on, every launcher brackets its kernel with a pair
// of HIP events on the stream it launches on; the report adds up the elapsed times per kernel.  Off (the default) the
// scopes cost one pointer test.  One recorder per process at a time: measurement aid, not part of the data path.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <vector>

namespace dafs {

enum stage_id {
  ST_CONTRAFOLD = 0, ST_CF_POSTERIOR, ST_BP_COMPACT, ST_PAIRHMM3, ST_PAIRHMM5, ST_MP_SIM, ST_MP_INTERLEAVE, ST_PCT_ROWS, ST_PCT_EMIT,
  ST_PCT_BP_ROWS, ST_PCT_BP_EMIT, ST_FOURWAY_ROWS, ST_NODE_AVG, ST_NODE_LISTS, ST_NODE_CBP_FILL, ST_DD_SOLVE, ST_NODE_PACK,
  ST_NUSSINOV_SINGLE, ST_NW_SINGLE, ST_COUNT
};
static const char* const kStageNames[ST_COUNT] = {
  "k_contrafold", "k_contrafold_posterior", "k_bp_compact", "k_pairhmm3", "k_pairhmm5", "k_mp_sim", "k_mp_interleave", "k_pct_rows", "k_pct_emit",
  "k_pct_bp_rows", "k_pct_bp_emit", "k_fourway_rows", "k_node_avg", "k_node_lists", "k_node_cbp_fill", "k_dd_solve", "k_node_pack",
  "k_nussinov_single", "k_nw_single"};

struct stage_recorder {
  struct rec { int id; hipEvent_t a, b; };
  std::vector<rec> recs;
  std::vector<hipEvent_t> pool;  // events of earlier reports, reused
  hipEvent_t take() {
    if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
  }
  void clear() { for (rec& r : recs) { pool.push_back(r.a); pool.push_back(r.b); } recs.clear(); }
  void release() { clear(); for (hipEvent_t e : pool) if (e) (void)hipEventDestroy(e); pool.clear(); }
};

stage_recorder*& stage_current();  // capi.cpp: the recorder that is switched on, or null

struct stage_scope {
  stage_recorder* r;
  hipStream_t st;
  hipEvent_t b = nullptr;
  bool go = true;  // STAGE_LAUNCH: the bracketed statement runs once
  stage_scope(int id, hipStream_t s) : r(stage_current()), st(s) {
    if (!r) return;
    hipEvent_t a = r->take();
    b = r->take();
    if (!a || !b) { r = nullptr; return; }
    (void)hipEventRecord(a, st);
    r->recs.push_back({id, a, b});
  }
  ~stage_scope() { if (r) (void)hipEventRecord(b, st); }
};

}  // namespace dafs

// brackets the next statement (a kernel launch on stream st) with the recorder's events
#define STAGE_LAUNCH(id, st) for (dafs::stage_scope stage_s_(id, st); stage_s_.go; stage_s_.go = false)
