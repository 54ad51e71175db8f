// dafs_amd/csrc/capi_pct.cpp -- L1: base-pairing store upload/fetch and the two probabilistic
// consistency transforms (reference src/dafs.cpp:258-375, called at :1822-1827).
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>

#include "../../include/dafs_hip.h"
#include "ctx.h"
#include "hip_util.h"
#include "pct.h"

using namespace dafs;

// AUXFold equivalent (reference src/fold.cpp:230-278): base-pairing probabilities supplied by
// the caller.  rowptr: per sequence len+1 entries (relative to that sequence's first entry),
// concatenated; col/val: all sequences' entries concatenated, rows ascending, j > i.
extern "C" int dafs_hip_set_bp(dafs_hip_ctx* c, const uint32_t* rowptr, const uint32_t* col, const float* val) {
  if (!c || c->len.empty() || !rowptr) return DAFS_HIP_EINVAL;
  if (hip_check(hipSetDevice(c->device))) return DAFS_HIP_ENODEV;
  const uint32_t n = (uint32_t)c->len.size();
  bp_store& st = c->bp[0];
  st.valid = false;
  c->bp[1].valid = false;
  c->cur_bp = 0;
  std::vector<uint64_t> bp_off(n + 1, 0);
  std::vector<uint32_t> nnz(n);
  for (uint32_t x = 0; x < n; ++x) {
    nnz[x] = rowptr[c->seq_rp_off[x] + c->len[x]];
    bp_off[x + 1] = bp_off[x] + nnz[x];
  }
  const uint64_t total = bp_off[n];
  if (total && (!col || !val)) return DAFS_HIP_EINVAL;
  int rc;
  if ((rc = st.rowptr.upload(rowptr, c->seq_rp_off[n], c->stream))) return rc;
  if ((rc = st.col.upload(col, total, c->stream))) return rc;
  if ((rc = st.val.upload(val, total, c->stream))) return rc;
  if ((rc = st.nnz.upload(nnz.data(), n, c->stream))) return rc;
  if ((rc = st.bp_off.upload(bp_off.data(), n + 1, c->stream))) return rc;
  if ((rc = st.rp_off.upload(c->seq_rp_off.data(), n + 1, c->stream))) return rc;
  st.total_nnz = total;
  st.valid = true;
  return DAFS_HIP_OK;
}

extern "C" int dafs_hip_bp_result_size(dafs_hip_ctx* c, int relaxed, uint64_t* total_nnz, uint64_t* total_rowptr) {
  if (!c || relaxed < 0 || relaxed > 1 || !c->bp[relaxed].valid) return DAFS_HIP_EINVAL;
  if (total_nnz) *total_nnz = c->bp[relaxed].total_nnz;
  if (total_rowptr) *total_rowptr = c->seq_rp_off.back();
  return DAFS_HIP_OK;
}

// Same layout as dafs_hip_set_bp, sequences in input order.
extern "C" int dafs_hip_bp_fetch(dafs_hip_ctx* c, int relaxed, uint32_t* rowptr, uint32_t* col, float* val) {
  if (!c || relaxed < 0 || relaxed > 1 || !c->bp[relaxed].valid) return DAFS_HIP_EINVAL;
  if (hip_check(hipSetDevice(c->device))) return DAFS_HIP_ENODEV;
  const bp_store& st = c->bp[relaxed];
  const uint32_t n = (uint32_t)c->len.size();
  int rc;
  if (rowptr && (rc = st.rowptr.download(rowptr, c->seq_rp_off[n]))) return rc;
  if (!col && !val) return DAFS_HIP_OK;
  std::vector<uint64_t> off(n + 1);
  std::vector<uint32_t> nnz(n);
  if ((rc = st.bp_off.download(off.data(), n + 1))) return rc;
  if ((rc = st.nnz.download(nnz.data(), n))) return rc;
  std::vector<uint32_t> h_col(st.total_nnz);
  std::vector<float> h_val(st.total_nnz);
  if ((rc = st.col.download(h_col.data(), st.total_nnz))) return rc;
  if ((rc = st.val.download(h_val.data(), st.total_nnz))) return rc;
  uint64_t w = 0;
  for (uint32_t x = 0; x < n; ++x) {  // the relaxed pool is bump-allocated: reorder by sequence
    if (col) memcpy(col + w, h_col.data() + off[x], nnz[x] * sizeof(uint32_t));
    if (val) memcpy(val + w, h_val.data() + off[x], nnz[x] * sizeof(float));
    w += nnz[x];
  }
  return DAFS_HIP_OK;
}

namespace dafs {
void pct_task_order(const uint32_t* pair_x, const uint32_t* pair_y, const uint32_t* len, uint32_t nseq, uint64_t p0, uint32_t count,
                    std::vector<uint2>& out) {
  // pairs of the launch bucketed by y (stable: x ascending within a bucket)
  std::vector<uint32_t> first(nseq + 1, 0), by_y(count);
  for (uint32_t k = 0; k < count; ++k) ++first[pair_y[p0 + k] + 1];
  for (uint32_t y = 0; y < nseq; ++y) first[y + 1] += first[y];
  {
    std::vector<uint32_t> cur(first.begin(), first.end() - 1);
    for (uint32_t k = 0; k < count; ++k) by_y[cur[pair_y[p0 + k]]++] = k;
  }
  std::vector<uint2> tasks;
  tasks.reserve((size_t)count * 12);
  for (uint32_t y = 0; y < nseq; ++y) {
    uint32_t rbmax = 0;
    for (uint32_t q = first[y]; q < first[y + 1]; ++q) rbmax = std::max(rbmax, (len[pair_x[p0 + by_y[q]]] + 15u) / 16u);
    for (uint32_t rb = 0; rb < rbmax; ++rb)
      for (uint32_t q = first[y]; q < first[y + 1]; ++q) {
        const uint32_t k = by_y[q];
        if (rb * 16u < len[pair_x[p0 + k]]) tasks.push_back(make_uint2(k, rb));
      }
  }
  const size_t total = tasks.size(), per = (total + 7) / 8;
  out.assign(per * 8, make_uint2(0xFFFFFFFFu, 0u));
  for (size_t b = 0; b < per * 8; ++b) {  // workgroup b runs on XCD b mod 8: each XCD gets one contiguous range of the sorted tasks
    const size_t src = (b % 8) * per + b / 8;
    if (src < total) out[b] = tasks[src];
  }
}
}  // namespace dafs

// relax_basepairing_probability then relax_matching_probability, both from the un-relaxed
// stores (dafs.cpp:1822-1827).  A weight of 0 skips that transform, as the reference does.
// which: bit 0 the base-pairing transform, bit 1 the matching transform (the two read only un-relaxed stores, so they
// may be run in either order, e.g. the matching transform while the folding kernels are still busy)
// fourway: the matching part runs DAFS::relax_fourway_consistency (weight w_pct_a) instead, and its result becomes the
// un-relaxed store (dafs_hip_fourway_consistency below)
static int consistency_parts(dafs_hip_ctx* c, float w_pct_a, float w_pct_s, int which, uint64_t pair_begin = 0, uint64_t pair_end = 0,
                             bool fourway = false) {
  if (!c || c->len.empty()) return DAFS_HIP_EINVAL;
  if (hip_check(hipSetDevice(c->device))) return DAFS_HIP_ENODEV;
  const uint32_t n = (uint32_t)c->len.size();
  const uint64_t all = (uint64_t)n * (n - 1) / 2;
  mp_store& raw = c->mp[0];
  if (!raw.valid || raw.n_tasks != all || (c->sim.empty() && !fourway)) return DAFS_HIP_EINVAL;
  if (fourway && !c->bp[0].valid) return DAFS_HIP_EINVAL;
  const uint32_t max_len = c->max_len();
  mp_store_dev mpv = raw.view(c->d_len.ptr, n);
  int rc;
  // the row kernels gather {column, value} pairs: one interleaved copy of the input entries per transform
  if ((rc = c->mp_ent2.reserve(raw.pool_used + 32))) return rc;  // + slack: k_pct_rows' unpredicated fetch reads up to 15 entries behind a row
  if ((rc = pct_interleave_launch(raw.col.ptr, raw.val.ptr, c->mp_ent2.ptr, raw.pool_used, c->stream))) return rc;
  mpv.ent2 = c->mp_ent2.ptr;
  {  // the identity matrix mp[x][x] as an ordinary CSR (entries {k, 1}, row pointers k): the row kernels treat it like any other
    const uint32_t ni = max_len + 1 + 32;
    if ((rc = c->mp_ident2.reserve((size_t)ni + ni / 2 + 2))) return rc;
    uint32_t* ident_rp = (uint32_t*)(c->mp_ident2.ptr + ni);
    if ((rc = pct_ident_launch(c->mp_ident2.ptr, ident_rp, ni, c->stream))) return rc;
    mpv.ident2 = c->mp_ident2.ptr;
    mpv.ident_rp = ident_rp;
  }
  if ((rc = c->counters.reserve(4))) return rc;
  if (pair_end == 0) pair_end = all;
  if (pair_begin > pair_end || pair_end > all) return DAFS_HIP_EINVAL;
  const bool shard = pair_begin != 0 || pair_end != all;  // only these output pairs are computed, the others stay empty

  if (which & 1) c->cur_bp = 0;
  if ((which & 1) && w_pct_s != 0.0f) {
    if (!c->bp[0].valid) return DAFS_HIP_EINVAL;
    bp_store& in = c->bp[0];
    bp_store& out = c->bp[1];
    out.valid = false;
    if ((rc = out.rowptr.reserve(c->seq_rp_off[n]))) return rc;
    if ((rc = out.nnz.reserve(n))) return rc;
    if ((rc = out.bp_off.reserve(n + 1))) return rc;
    if ((rc = out.rp_off.upload(c->seq_rp_off.data(), n + 1, c->stream))) return rc;
    uint64_t cap = std::max<uint64_t>(4 * in.total_nnz + 1024, 16ull * c->off[n]);
    for (int attempt = 0;; ++attempt) {
      if ((rc = out.col.reserve(cap))) return rc;
      if ((rc = out.val.reserve(cap))) return rc;
      if (hip_check(hipMemsetAsync(c->counters.ptr, 0, 4 * sizeof(unsigned long long), c->stream))) return DAFS_HIP_ELAUNCH;
      pct_bp_args a;
      memset(&a, 0, sizeof a);
      a.mp = mpv; a.bp = in.view(); a.sim = c->d_sim.ptr; a.w_pct = w_pct_s;
      a.out_rowptr = out.rowptr.ptr; a.out_col = out.col.ptr; a.out_val = out.val.ptr;
      a.pool_top = c->counters.ptr; a.pool_cap = cap; a.out_off = out.bp_off.ptr; a.out_nnz = out.nnz.ptr;
      a.status = (int*)(c->counters.ptr + 2);
      {  // dense tiles, one per sequence
        std::vector<uint64_t> toff(n);
        uint64_t cells = 0;
        for (uint32_t x = 0; x < n; ++x) { toff[x] = cells; cells += (uint64_t)c->len[x] * c->len[x]; }
        if ((rc = c->scratch.reserve(cells + 64))) return rc;
        if ((rc = c->work.reserve((size_t)n * 8 + 256))) return rc;
        if ((rc = c->work2.reserve((size_t)n * 4 + 256))) return rc;
        if (hip_check(hipMemcpyAsync(c->work.ptr, toff.data(), (size_t)n * 8, hipMemcpyHostToDevice, c->stream))) return DAFS_HIP_ELAUNCH;
        if (hip_check(hipStreamSynchronize(c->stream))) return DAFS_HIP_ELAUNCH;
        a.tile = c->scratch.ptr; a.tile_off = (const uint64_t*)c->work.ptr; a.sum_w = (float*)c->work2.ptr;
      }
      if ((rc = pct_bp_launch(a, max_len, c->stream))) return rc;
      unsigned long long h[4];
      if (hip_check(hipMemcpyAsync(h, c->counters.ptr, sizeof h, hipMemcpyDeviceToHost, c->stream))) return DAFS_HIP_ELAUNCH;
      if (hip_check(hipStreamSynchronize(c->stream))) return DAFS_HIP_ELAUNCH;
      const int status = (int)(h[2] & 0xffffffffu);
      if (status == 0) { out.total_nnz = h[0]; out.valid = true; break; }
      if (status != DAFS_HIP_EOVERFLOW || attempt >= 4) return status;
      cap = std::max<uint64_t>(h[0], cap * 2);
    }
    c->cur_bp = 1;
  }

  if (which & 2) c->cur_mp = 0;
  if ((which & 2) && w_pct_a != 0.0f) {
    mp_store& out = c->mp[1];
    out.valid = false;
    out.n_tasks = all;
    out.pair_x = raw.pair_x;
    out.pair_y = raw.pair_y;
    out.rp_by_pair = raw.rp_by_pair;
    out.rp_total = raw.rp_total;
    out.task_of_pair.resize(all);
    for (uint64_t p = 0; p < all; ++p) out.task_of_pair[p] = (uint32_t)p;
    if ((rc = out.d_task_of_pair.upload(out.task_of_pair.data(), all, c->stream))) return rc;
    if ((rc = out.rp_off.upload(out.rp_by_pair.data(), all, c->stream))) return rc;
    if ((rc = out.rowptr_pool.reserve(out.rp_total))) return rc;
    if ((rc = out.pair_off.reserve(all))) return rc;
    if ((rc = out.pair_nnz.reserve(all))) return rc;
    if ((rc = c->d_pair_x.upload(raw.pair_x.data(), all, c->stream))) return rc;
    if ((rc = c->d_pair_y.upload(raw.pair_y.data(), all, c->stream))) return rc;
    uint64_t cap = std::max<uint64_t>(raw.pool_used * 2 + 1024, out.pool_cap_hint);
    for (int attempt = 0;; ++attempt) {
      if ((rc = out.col.reserve(cap))) return rc;
      if ((rc = out.val.reserve(cap))) return rc;
      if (hip_check(hipMemsetAsync(c->counters.ptr, 0, 4 * sizeof(unsigned long long), c->stream))) return DAFS_HIP_ELAUNCH;
      if (shard) {  // pairs outside the shard: no entries, all row pointers 0
        if (hip_check(hipMemsetAsync(out.rowptr_pool.ptr, 0, out.rp_total * sizeof(uint32_t), c->stream))) return DAFS_HIP_ELAUNCH;
        if (hip_check(hipMemsetAsync(out.pair_off.ptr, 0, all * sizeof(uint64_t), c->stream))) return DAFS_HIP_ELAUNCH;
        if (hip_check(hipMemsetAsync(out.pair_nnz.ptr, 0, all * sizeof(uint32_t), c->stream))) return DAFS_HIP_ELAUNCH;
      }
      pct_match_args a;
      memset(&a, 0, sizeof a);
      a.in = mpv; a.sim = c->d_sim.ptr; a.pair_x = c->d_pair_x.ptr; a.pair_y = c->d_pair_y.ptr;
      a.npairs = (uint32_t)all; a.w_pct = w_pct_a;
      a.rp_off = out.rp_off.ptr; a.rowptr_pool = out.rowptr_pool.ptr; a.col = out.col.ptr; a.val = out.val.ptr;
      a.pool_top = c->counters.ptr; a.pool_cap = cap; a.pair_off = out.pair_off.ptr; a.pair_nnz = out.pair_nnz.ptr;
      a.status = (int*)(c->counters.ptr + 2);
      if (fourway) { a.bp = c->bp[0].view(); a.w_f = w_pct_a; }
      // dense row tiles, in batches of at most kTileFloats; the tile memory is the pair kernels' scratch
      const uint64_t kTileFloats = 1ull << 31;  // 8 GiB
      for (uint64_t p0 = pair_begin; p0 < pair_end;) {
        std::vector<uint64_t> toff;
        uint64_t cells = 0, p1 = p0;
        while (p1 < pair_end) {
          const uint64_t need = (uint64_t)c->len[raw.pair_x[p1]] * c->len[raw.pair_y[p1]];
          if (p1 > p0 && cells + need > kTileFloats) break;
          toff.push_back(cells);
          cells += need;
          ++p1;
        }
        if ((rc = c->scratch.reserve(cells + 64))) return rc;
        const size_t cnt = (size_t)(p1 - p0);
        if ((rc = c->work.reserve(cnt * 8 + 256))) return rc;
        if ((rc = c->work2.reserve(cnt * 4 + 256))) return rc;
        if (hip_check(hipMemcpyAsync(c->work.ptr, toff.data(), cnt * 8, hipMemcpyHostToDevice, c->stream))) return DAFS_HIP_ELAUNCH;
        if (hip_check(hipStreamSynchronize(c->stream))) return DAFS_HIP_ELAUNCH;  // toff dies at the end of the scope
        a.tile = c->scratch.ptr; a.tile_off = (const uint64_t*)c->work.ptr; a.sum_w = (float*)c->work2.ptr;
        a.wg_task = nullptr; a.wg_tasks = 0;
        // DAFS_HIP_PCT_YORDER=1 (tuning aid): workgroups ordered by (y, row block, x) and dealt to the XCDs in contiguous ranges.
        // Measured slower than the plain 2-D grid (N=128: 13.3 against 12.8 ms for the stage, N=256: 135 against 120): the
        // gathers are not what binds the kernel, so the default stays the 2-D grid.
        if (!fourway && getenv("DAFS_HIP_PCT_YORDER")) {
          std::vector<uint2> order;
          pct_task_order(raw.pair_x.data(), raw.pair_y.data(), c->len.data(), n, p0, (uint32_t)cnt, order);
          if ((rc = c->pct_tasks.upload(order.data(), order.size(), c->stream))) return rc;  // synchronises
          a.wg_task = c->pct_tasks.ptr; a.wg_tasks = (uint32_t)order.size();
        }
        if ((rc = fourway ? pct_fourway_launch(a, max_len, (uint32_t)p0, (uint32_t)cnt, c->stream)
                          : pct_match_launch(a, max_len, (uint32_t)p0, (uint32_t)cnt, c->stream))) return rc;
        p0 = p1;
      }
      unsigned long long h[4];
      if (hip_check(hipMemcpyAsync(h, c->counters.ptr, sizeof h, hipMemcpyDeviceToHost, c->stream))) return DAFS_HIP_ELAUNCH;
      if (hip_check(hipStreamSynchronize(c->stream))) return DAFS_HIP_ELAUNCH;
      const int status = (int)(h[2] & 0xffffffffu);
      if (status == 0) { out.pool_used = h[0]; out.pool_cap_hint = cap; out.valid = true; break; }
      if (status != DAFS_HIP_EOVERFLOW || attempt >= 4) return status;
      cap = std::max<uint64_t>(h[0], cap * 2);
    }
    c->cur_mp = 1;
  }
  return DAFS_HIP_OK;
}

// DAFS::relax_fourway_consistency (dafs.cpp:377-444, called at :1808-1809 when -f is not 0): mp_ is replaced by its mix with
// the stacking evidence of the base-pairing matrices, BEFORE the similarity scores are taken -- so the result becomes the
// context's un-relaxed store and the scores are recomputed from it.  Needs the un-relaxed base-pairing store.
extern "C" int dafs_hip_fourway_consistency(dafs_hip_ctx* c, float w_pct_f) {
  if (w_pct_f == 0.0f) return DAFS_HIP_OK;
  int rc = consistency_parts(c, w_pct_f, 0.0f, 2, 0, 0, true);
  if (rc) return rc;
  std::swap(c->mp[0], c->mp[1]);
  c->mp[1].valid = false;
  c->cur_mp = 0;
  return dafs_recompute_sim(c, c->mp[0]);
}

extern "C" int dafs_hip_consistency(dafs_hip_ctx* c, float w_pct_a, float w_pct_s) { return consistency_parts(c, w_pct_a, w_pct_s, 3); }
extern "C" int dafs_hip_consistency_match(dafs_hip_ctx* c, float w_pct_a) { return consistency_parts(c, w_pct_a, 0.0f, 2); }
extern "C" int dafs_hip_consistency_bp(dafs_hip_ctx* c, float w_pct_s) { return consistency_parts(c, 0.0f, w_pct_s, 1); }
// The matching transform for the output pairs [pair_begin, pair_end) only (row-major pair index; every output pair
// reads all the un-relaxed matrices, none reads another output: dafs.cpp:265-315).  The other pairs of the relaxed
// store stay empty; a multi-GPU run gathers the shards and installs the whole with dafs_hip_mp_install.
extern "C" int dafs_hip_consistency_match_range(dafs_hip_ctx* c, float w_pct_a, uint64_t pair_begin, uint64_t pair_end) {
  if (w_pct_a == 0.0f) return DAFS_HIP_EINVAL;
  return consistency_parts(c, w_pct_a, 0.0f, 2, pair_begin, pair_end);
}
