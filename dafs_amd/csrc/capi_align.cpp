// dafs_amd/csrc/capi_align.cpp -- L1: batch alignment posteriors (Align::Model::calculate,
// reference src/align.cpp:35-52) and access to the matching-probability stores.
#include <hip/hip_runtime.h>
#include <string.h>
#include <algorithm>
#include <numeric>
#include <vector>

#include "../../include/dafs_hip.h"
#include "ctx.h"
#include "pct.h"
#include "hip_util.h"

using namespace dafs;

// pair index p <-> (i<j), row-major as in Align::Model::calculate (align.cpp:39-50)
static void pair_from_index(uint64_t p, uint32_t n, uint32_t* i, uint32_t* j) {
  uint32_t a = 0;
  uint64_t rem = p;
  while (rem >= (uint64_t)(n - 1 - a)) { rem -= (n - 1 - a); ++a; }
  *i = a;
  *j = a + 1 + (uint32_t)rem;
}

extern "C" int dafs_hip_align_posteriors(dafs_hip_ctx* c, int model, float th, uint64_t pair_begin, uint64_t pair_end) {
  if (!c || c->len.empty() || !(th >= 0.0f)) return DAFS_HIP_EINVAL;
  if (model != DAFS_ALIGN_PROBCONS && model != DAFS_ALIGN_CONTRALIGN) return DAFS_HIP_EINVAL;
  if (hip_check(hipSetDevice(c->device))) return DAFS_HIP_ENODEV;
  const uint32_t n = (uint32_t)c->len.size();
  const uint64_t all = (uint64_t)n * (n - 1) / 2;
  if (pair_end == 0) pair_end = all;
  if (pair_begin > pair_end || pair_end > all) return DAFS_HIP_EINVAL;
  const uint64_t np = pair_end - pair_begin;
  mp_store& st = c->mp[0];
  st.valid = false;
  c->mp[1].valid = false;
  c->cur_mp = 0;
  c->sim.clear();
  st.pair_x.resize(np);
  st.pair_y.resize(np);
  st.n_tasks = np;
  if (np == 0) { st.valid = true; st.rp_total = st.pool_used = 0; return DAFS_HIP_OK; }
  {
    uint32_t i, j;
    pair_from_index(pair_begin, n, &i, &j);
    for (uint64_t p = 0; p < np; ++p) {
      st.pair_x[p] = i;
      st.pair_y[p] = j;
      if (++j == n) { ++i; j = i + 1; }
    }
  }
  // processing order: longest first (cost ~ len1*len2), so the work queue balances the tail
  std::vector<uint32_t> order(np);
  std::iota(order.begin(), order.end(), 0u);
  std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) {
    const uint64_t ca = (uint64_t)c->len[st.pair_x[a]] * c->len[st.pair_y[a]];
    const uint64_t cb = (uint64_t)c->len[st.pair_x[b]] * c->len[st.pair_y[b]];
    return ca > cb;
  });
  std::vector<dafs_pair_task> tasks(np);
  std::vector<uint64_t> rp_off(np);
  uint32_t max1 = 0, max2 = 0;
  uint64_t rp_total = 0, est = 0;
  // row pointers are laid out in shard order so fetch can copy them out unchanged
  st.rp_by_pair.resize(np);
  for (uint64_t p = 0; p < np; ++p) {
    st.rp_by_pair[p] = rp_total;
    rp_total += (uint64_t)c->len[st.pair_x[p]] + 1 + c->len[st.pair_y[p]] + 1;
  }
  st.task_of_pair.resize(np);
  for (uint64_t k = 0; k < np; ++k) {
    const uint32_t p = order[k];
    const uint32_t x = st.pair_x[p], y = st.pair_y[p];
    st.task_of_pair[p] = (uint32_t)k;
    tasks[k] = {c->off[x], c->len[x], c->off[y], c->len[y]};
    rp_off[k] = st.rp_by_pair[p];
    max1 = std::max(max1, c->len[x]);
    max2 = std::max(max2, c->len[y]);
    est += 2ull * std::min(c->len[x], c->len[y]) * 24;
  }
  dafs_pairhmm_plan plan;
  int rc = model == DAFS_ALIGN_PROBCONS ? dafs_hipk_pairhmm_plan((uint32_t)np, max1, max2, &plan)
                                        : dafs_hipk_pairhmm5_plan((uint32_t)np, max1, max2, &plan);
  if (rc) return rc;
  st.rp_total = rp_total;
  if ((rc = c->tasks.upload(tasks.data(), np, c->stream))) return rc;
  if ((rc = st.rp_off.upload(rp_off.data(), np, c->stream))) return rc;
  if ((rc = st.d_task_of_pair.upload(st.task_of_pair.data(), np, c->stream))) return rc;
  if ((rc = c->scratch.reserve(plan.scratch_bytes / sizeof(float)))) return rc;
  if ((rc = st.rowptr_pool.reserve(rp_total))) return rc;
  if ((rc = st.pair_off.reserve(np))) return rc;
  if ((rc = st.pair_nnz.reserve(np))) return rc;
  if ((rc = c->task_sim.reserve(np))) return rc;
  if ((rc = c->counters.reserve(4))) return rc;
  uint64_t cap = std::max<uint64_t>(est, 1024);
  if (st.pool_cap_hint > cap) cap = st.pool_cap_hint;

  for (int attempt = 0; attempt < 6; ++attempt) {
    if ((rc = st.col.reserve(cap))) return rc;
    if ((rc = st.val.reserve(cap))) return rc;
    if (hip_check(hipMemsetAsync(c->counters.ptr, 0, 4 * sizeof(unsigned long long), c->stream))) return DAFS_HIP_ELAUNCH;
    auto fill = [&](auto& a) {
      memset(&a, 0, sizeof a);
      a.codes = c->codes.ptr;
      a.tasks = c->tasks.ptr;
      a.ntasks = (uint32_t)np;
      a.th = th;
      a.scratch = c->scratch.ptr;
      a.queue = (uint32_t*)(c->counters.ptr + 1);
      a.rp_off = st.rp_off.ptr;
      a.rowptr_pool = st.rowptr_pool.ptr;
      a.ent_col = st.col.ptr;
      a.ent_val = st.val.ptr;
      a.pool_top = c->counters.ptr;
      a.pool_cap = cap;
      a.pair_off = st.pair_off.ptr;
      a.pair_nnz = st.pair_nnz.ptr;
      a.sim = c->task_sim.ptr;
      a.status = (int*)(c->counters.ptr + 2);
    };
    if (model == DAFS_ALIGN_PROBCONS) {
      dafs_pairhmm3_args a;
      fill(a);
      dafs_hip_pairhmm3_default_model(&a.model);
      if ((rc = dafs_hipk_pairhmm3_launch(&a, &plan, c->stream))) return rc;
    } else {
      dafs_pairhmm5_args a;
      fill(a);
      dafs_hip_pairhmm5_default_model(&a.model);
      if ((rc = dafs_hipk_pairhmm5_launch(&a, &plan, c->stream))) return rc;
    }
    unsigned long long host_cnt[4];
    if (hip_check(hipMemcpyAsync(host_cnt, c->counters.ptr, sizeof host_cnt, hipMemcpyDeviceToHost, c->stream))) return DAFS_HIP_ELAUNCH;
    if (hip_check(hipStreamSynchronize(c->stream))) return DAFS_HIP_ELAUNCH;
    const int status = (int)(host_cnt[2] & 0xffffffffu);
    if (status == 0) {
      st.pool_used = host_cnt[0];
      st.pool_cap_hint = cap;
      st.valid = true;
      c->plan = plan;
      // similarity matrix (dafs.cpp:1813-1819) when the shard is the whole pair set
      if (np == all) {
        std::vector<float> ts(np);
        if ((rc = c->task_sim.download(ts.data(), np))) return rc;
        c->sim.assign((size_t)n * n, 0.0f);
        for (uint32_t i = 0; i < n; ++i) c->sim[(size_t)i * n + i] = 1.0f;
        for (uint64_t p = 0; p < np; ++p) {
          const float s = ts[st.task_of_pair[p]];
          c->sim[(size_t)st.pair_x[p] * n + st.pair_y[p]] = s;
          c->sim[(size_t)st.pair_y[p] * n + st.pair_x[p]] = s;
        }
        if ((rc = c->d_sim.upload(c->sim.data(), c->sim.size(), c->stream))) return rc;
      }
      return DAFS_HIP_OK;
    }
    if (status != DAFS_HIP_EOVERFLOW) return status;
    cap = std::max<uint64_t>(host_cnt[0], cap * 2);  // pool_top kept counting: exact requirement
  }
  return DAFS_HIP_EOVERFLOW;
}

// AUXAlign::calculate (src/align.cpp:204-246, --align-aux) and the hand-over point after an all-gather of
// shards: the caller supplies the rows of mp[x][y] for every pair x < y (row-major pair order); the
// transposes (transpose_mp, dafs.cpp:155-167) are laid out here and the similarity scores
// (calculate_similarity_score, :713-764, :1813-1819) are computed on the device.
extern "C" int dafs_hip_set_mp(dafs_hip_ctx* c, const uint32_t* nnz, const uint32_t* rowptr, const uint32_t* col, const float* val) {
  if (!c || c->len.size() < 2 || !nnz || !rowptr || !col || !val) return DAFS_HIP_EINVAL;
  if (hip_check(hipSetDevice(c->device))) return DAFS_HIP_ENODEV;
  const uint32_t n = (uint32_t)c->len.size();
  const uint64_t np = (uint64_t)n * (n - 1) / 2;
  mp_store& st = c->mp[0];
  st.valid = false;
  c->mp[1].valid = false;
  c->cur_mp = 0;
  c->sim.clear();
  st.pair_x.resize(np); st.pair_y.resize(np); st.task_of_pair.resize(np); st.rp_by_pair.resize(np);
  st.n_tasks = np;
  std::vector<uint64_t> pair_off(np);
  uint64_t rp_total = 0, ent_total = 0, p = 0;
  for (uint32_t x = 0; x < n; ++x)
    for (uint32_t y = x + 1; y < n; ++y, ++p) {
      st.pair_x[p] = x; st.pair_y[p] = y; st.task_of_pair[p] = (uint32_t)p;
      st.rp_by_pair[p] = rp_total;
      pair_off[p] = ent_total;
      rp_total += (uint64_t)c->len[x] + 1 + c->len[y] + 1;
      ent_total += 2ull * nnz[p];
    }
  std::vector<uint32_t> h_rp(rp_total), h_col(ent_total + 1);
  std::vector<float> h_val(ent_total + 1);
  uint64_t rin = 0, ein = 0;
  std::vector<uint32_t> cnt;
  for (p = 0; p < np; ++p) {
    const uint32_t L1 = c->len[st.pair_x[p]], L2 = c->len[st.pair_y[p]], m = nnz[p];
    const uint32_t* rp = rowptr + rin;
    if (rp[0] != 0 || rp[L1] != m) return DAFS_HIP_EINVAL;
    uint32_t* orp = h_rp.data() + st.rp_by_pair[p];
    uint32_t* ocol = h_col.data() + pair_off[p];
    float* oval = h_val.data() + pair_off[p];
    // rows of mp[x][y] as given (columns ascending within a row)
    for (uint32_t i = 0; i <= L1; ++i) orp[i] = rp[i];
    cnt.assign((size_t)L2 + 1, 0);
    for (uint32_t i = 0; i < L1; ++i) {
      if (rp[i + 1] < rp[i] || rp[i + 1] > m) return DAFS_HIP_EINVAL;
      for (uint32_t e = rp[i]; e < rp[i + 1]; ++e) {
        const uint32_t j = col[ein + e];
        if (j >= L2 || (e > rp[i] && col[ein + e - 1] >= j)) return DAFS_HIP_EINVAL;
        ocol[e] = j; oval[e] = val[ein + e];
        ++cnt[j + 1];
      }
    }
    // rows of mp[y][x]: a counting sort by column keeps the rows of one column in ascending order
    uint32_t* trp = orp + L1 + 1;
    trp[0] = 0;
    for (uint32_t j = 0; j < L2; ++j) trp[j + 1] = trp[j] + cnt[j + 1];
    for (uint32_t j = 0; j < L2; ++j) cnt[j] = trp[j];
    for (uint32_t i = 0; i < L1; ++i)
      for (uint32_t e = rp[i]; e < rp[i + 1]; ++e) {
        const uint32_t w = cnt[col[ein + e]]++;
        ocol[m + w] = i; oval[m + w] = val[ein + e];
      }
    rin += (uint64_t)L1 + 1;
    ein += m;
  }
  int rc;
  st.rp_total = rp_total;
  st.pool_used = ent_total;
  st.pool_cap_hint = std::max<uint64_t>(st.pool_cap_hint, ent_total);
  std::vector<uint32_t> h_nnz(nnz, nnz + np);
  if ((rc = st.rowptr_pool.upload(h_rp.data(), rp_total, c->stream))) return rc;
  if ((rc = st.col.upload(h_col.data(), ent_total + 1, c->stream))) return rc;
  if ((rc = st.val.upload(h_val.data(), ent_total + 1, c->stream))) return rc;
  if ((rc = st.pair_off.upload(pair_off.data(), np, c->stream))) return rc;
  if ((rc = st.pair_nnz.upload(h_nnz.data(), np, c->stream))) return rc;
  if ((rc = st.rp_off.upload(st.rp_by_pair.data(), np, c->stream))) return rc;
  if ((rc = st.d_task_of_pair.upload(st.task_of_pair.data(), np, c->stream))) return rc;
  if ((rc = c->d_pair_x.upload(st.pair_x.data(), np, c->stream))) return rc;
  if ((rc = c->d_pair_y.upload(st.pair_y.data(), np, c->stream))) return rc;
  if ((rc = dafs_recompute_sim(c, st))) return rc;
  st.valid = true;
  return DAFS_HIP_OK;
}

// Similarity scores (calculate_similarity_score, dafs.cpp:713-764, :1813-1819) of every pair from the rows of a store
// whose pairs are in row-major order (task == pair): device DP, then the host and device copies of sim.
int dafs_recompute_sim(dafs_hip_ctx* c, dafs::mp_store& st) {
  const uint32_t n = (uint32_t)c->len.size();
  const uint64_t np = (uint64_t)n * (n - 1) / 2;
  int rc;
  if ((rc = c->d_pair_x.upload(st.pair_x.data(), np, c->stream))) return rc;
  if ((rc = c->d_pair_y.upload(st.pair_y.data(), np, c->stream))) return rc;
  const uint32_t max_len = c->max_len();
  if ((rc = c->task_sim.reserve(np))) return rc;
  if ((rc = c->scratch.reserve(2 * ((size_t)max_len + 1) * np))) return rc;
  float* row_dp = c->scratch.ptr;
  int* row_tr = (int*)(c->scratch.ptr + ((size_t)max_len + 1) * np);
  if ((rc = mp_sim_launch(st.view(c->d_len.ptr, n), c->d_pair_x.ptr, c->d_pair_y.ptr, (uint32_t)np, c->task_sim.ptr, row_dp, row_tr, c->stream))) return rc;
  std::vector<float> ts(np);
  if (hip_check(hipStreamSynchronize(c->stream))) return DAFS_HIP_ELAUNCH;
  if ((rc = c->task_sim.download(ts.data(), np))) return rc;
  c->sim.assign((size_t)n * n, 0.0f);
  for (uint32_t i = 0; i < n; ++i) c->sim[(size_t)i * n + i] = 1.0f;
  for (uint64_t p = 0; p < np; ++p) {
    c->sim[(size_t)st.pair_x[p] * n + st.pair_y[p]] = ts[p];
    c->sim[(size_t)st.pair_y[p] * n + st.pair_x[p]] = ts[p];
  }
  return c->d_sim.upload(c->sim.data(), c->sim.size(), c->stream);
}

// A complete matching-probability store from arrays in the layout dafs_hip_mp_fetch / dafs_hip_align_fetch write (all
// pairs x < y in row-major order; per pair len_x+1 row pointers of mp[x][y] then len_y+1 of mp[y][x], relative to the
// pair; per pair nnz entries of mp[x][y] then nnz of mp[y][x]): what the ranks of a multi-GPU run hold after gathering
// their shards.  relaxed = 0 installs the models' posteriors together with the similarity scores sim[npairs]
// (calculate_similarity_score of every pair, computed by the shard's kernel); relaxed = 1 installs the result of the
// consistency transform on top of an installed or computed un-relaxed store.  Nothing is recomputed: uploads only.
extern "C" int dafs_hip_mp_install(dafs_hip_ctx* c, int relaxed, const uint32_t* nnz, const uint32_t* rowptr, const uint32_t* col,
                                   const float* val, const float* sim) {
  if (!c || c->len.size() < 2 || relaxed < 0 || relaxed > 1 || !nnz || !rowptr) return DAFS_HIP_EINVAL;
  if (relaxed == 0 && !sim) return DAFS_HIP_EINVAL;
  if (relaxed == 1 && (!c->mp[0].valid || c->sim.empty())) return DAFS_HIP_EINVAL;
  if (hip_check(hipSetDevice(c->device))) return DAFS_HIP_ENODEV;
  const uint32_t n = (uint32_t)c->len.size();
  const uint64_t np = (uint64_t)n * (n - 1) / 2;
  mp_store& st = c->mp[relaxed];
  st.valid = false;
  if (relaxed == 0) { c->mp[1].valid = false; c->cur_mp = 0; c->sim.clear(); }
  st.pair_x.resize(np); st.pair_y.resize(np); st.task_of_pair.resize(np); st.rp_by_pair.resize(np);
  st.n_tasks = np;
  std::vector<uint64_t> pair_off(np);
  uint64_t rp_total = 0, ent_total = 0, p = 0;
  for (uint32_t x = 0; x < n; ++x)
    for (uint32_t y = x + 1; y < n; ++y, ++p) {
      st.pair_x[p] = x; st.pair_y[p] = y; st.task_of_pair[p] = (uint32_t)p;
      st.rp_by_pair[p] = rp_total;
      pair_off[p] = ent_total;
      const uint32_t* rp = rowptr + rp_total;
      if (rp[0] != 0 || rp[c->len[x]] != nnz[p] || rp[c->len[x] + 1] != 0 || rp[c->len[x] + 1 + c->len[y]] != nnz[p]) return DAFS_HIP_EINVAL;
      rp_total += (uint64_t)c->len[x] + 1 + c->len[y] + 1;
      ent_total += 2ull * nnz[p];
    }
  if (ent_total && (!col || !val)) return DAFS_HIP_EINVAL;
  int rc;
  st.rp_total = rp_total;
  st.pool_used = ent_total;
  st.pool_cap_hint = std::max<uint64_t>(st.pool_cap_hint, ent_total);
  if ((rc = st.rowptr_pool.upload(rowptr, rp_total, c->stream))) return rc;
  if ((rc = st.col.reserve(ent_total + 1))) return rc;
  if ((rc = st.val.reserve(ent_total + 1))) return rc;
  if (ent_total) {
    if ((rc = st.col.upload(col, ent_total, c->stream))) return rc;
    if ((rc = st.val.upload(val, ent_total, c->stream))) return rc;
  }
  if ((rc = st.pair_off.upload(pair_off.data(), np, c->stream))) return rc;
  if ((rc = st.pair_nnz.upload(nnz, np, c->stream))) return rc;
  if ((rc = st.rp_off.upload(st.rp_by_pair.data(), np, c->stream))) return rc;
  if ((rc = st.d_task_of_pair.upload(st.task_of_pair.data(), np, c->stream))) return rc;
  if ((rc = c->d_pair_x.upload(st.pair_x.data(), np, c->stream))) return rc;
  if ((rc = c->d_pair_y.upload(st.pair_y.data(), np, c->stream))) return rc;
  if (relaxed == 0) {
    c->sim.assign((size_t)n * n, 0.0f);
    for (uint32_t i = 0; i < n; ++i) c->sim[(size_t)i * n + i] = 1.0f;
    for (p = 0; p < np; ++p) {
      c->sim[(size_t)st.pair_x[p] * n + st.pair_y[p]] = sim[p];
      c->sim[(size_t)st.pair_y[p] * n + st.pair_x[p]] = sim[p];
    }
    if ((rc = c->d_sim.upload(c->sim.data(), c->sim.size(), c->stream))) return rc;
    if ((rc = c->task_sim.upload(sim, np, c->stream))) return rc;
  } else {
    c->cur_mp = 1;
  }
  st.valid = true;
  return DAFS_HIP_OK;
}

extern "C" int dafs_hip_mp_result_size(dafs_hip_ctx* c, int relaxed, uint64_t* npairs, uint64_t* total_nnz, uint64_t* total_rowptr) {
  if (!c || relaxed < 0 || relaxed > 1 || !c->mp[relaxed].valid) return DAFS_HIP_EINVAL;
  const mp_store& st = c->mp[relaxed];
  if (npairs) *npairs = st.n_tasks;
  if (total_nnz) *total_nnz = st.pool_used / 2;
  if (total_rowptr) *total_rowptr = st.rp_total;
  return DAFS_HIP_OK;
}

extern "C" int dafs_hip_mp_fetch(dafs_hip_ctx* c, int relaxed, uint32_t* pair_x, uint32_t* pair_y, uint32_t* nnz,
                                 uint32_t* rowptr, uint32_t* col, float* val) {
  if (!c || relaxed < 0 || relaxed > 1 || !c->mp[relaxed].valid) return DAFS_HIP_EINVAL;
  if (hip_check(hipSetDevice(c->device))) return DAFS_HIP_ENODEV;
  const mp_store& st = c->mp[relaxed];
  const uint64_t np = st.n_tasks;
  if (np == 0) return DAFS_HIP_OK;
  int rc;
  if (pair_x) memcpy(pair_x, st.pair_x.data(), np * sizeof(uint32_t));
  if (pair_y) memcpy(pair_y, st.pair_y.data(), np * sizeof(uint32_t));
  std::vector<uint32_t> h_nnz(np);
  std::vector<uint64_t> h_off(np);
  if ((rc = st.pair_nnz.download(h_nnz.data(), np))) return rc;
  if ((rc = st.pair_off.download(h_off.data(), np))) return rc;
  if (nnz) for (uint64_t p = 0; p < np; ++p) nnz[p] = h_nnz[st.task_of_pair[p]];
  if (rowptr && (rc = st.rowptr_pool.download(rowptr, st.rp_total))) return rc;
  if (col || val) {
    std::vector<uint32_t> h_col;
    std::vector<float> h_val;
    if (col) { h_col.resize(st.pool_used); if ((rc = st.col.download(h_col.data(), st.pool_used))) return rc; }
    if (val) { h_val.resize(st.pool_used); if ((rc = st.val.download(h_val.data(), st.pool_used))) return rc; }
    uint64_t w = 0;
    for (uint64_t p = 0; p < np; ++p) {
      const uint64_t k = st.task_of_pair[p];
      const uint64_t n2 = 2ull * h_nnz[k];
      if (col) memcpy(col + w, h_col.data() + h_off[k], n2 * sizeof(uint32_t));
      if (val) memcpy(val + w, h_val.data() + h_off[k], n2 * sizeof(float));
      w += n2;
    }
  }
  return DAFS_HIP_OK;
}

extern "C" int dafs_hip_align_result_size(dafs_hip_ctx* c, uint64_t* npairs, uint64_t* total_nnz, uint64_t* total_rowptr) {
  return dafs_hip_mp_result_size(c, 0, npairs, total_nnz, total_rowptr);
}

extern "C" int dafs_hip_align_fetch(dafs_hip_ctx* c, uint32_t* pair_x, uint32_t* pair_y, float* sim, uint32_t* nnz,
                                    uint32_t* rowptr, uint32_t* col, float* val) {
  int rc = dafs_hip_mp_fetch(c, 0, pair_x, pair_y, nnz, rowptr, col, val);
  if (rc || !sim) return rc;
  const mp_store& st = c->mp[0];
  std::vector<float> ts(st.n_tasks);
  if ((rc = c->task_sim.download(ts.data(), st.n_tasks))) return rc;
  for (uint64_t p = 0; p < st.n_tasks; ++p) sim[p] = ts[st.task_of_pair[p]];
  return DAFS_HIP_OK;
}

// sim_ (dafs.cpp:1813-1819), N*N with unit diagonal; needs a full-pair-set align_posteriors
extern "C" int dafs_hip_get_sim(dafs_hip_ctx* c, float* sim) {
  if (!c || !sim || c->sim.empty()) return DAFS_HIP_EINVAL;
  memcpy(sim, c->sim.data(), c->sim.size() * sizeof(float));
  return DAFS_HIP_OK;
}
