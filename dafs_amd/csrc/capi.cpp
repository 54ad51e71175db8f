// dafs_amd/csrc/capi.cpp -- L1 "plugin" layer of the C ABI (include/dafs_hip.h): host buffers in,
// host buffers out, device workspace owned by a context object.  No CPU fallback: every entry
// point needs a HIP device and reports DAFS_HIP_ENODEV otherwise.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <numeric>
#include <string>
#include <vector>

#include "../../include/dafs_hip.h"
#include "ctx.h"
#include "hip_util.h"

namespace dafs {

static thread_local std::string g_last_error;
bool hip_check(hipError_t e) {
  if (e == hipSuccess) return false;
  g_last_error = hipGetErrorString(e);
  return true;
}

}  // namespace dafs

using namespace dafs;

extern "C" const char* dafs_hip_last_error(void) { return g_last_error.c_str(); }

extern "C" const char* dafs_hip_strerror(int code) {
  switch (code) {
    case DAFS_HIP_OK: return "ok";
    case DAFS_HIP_EINVAL: return "dafs_hip: invalid argument";
    case DAFS_HIP_ENODEV: return "dafs_hip: no usable HIP device";
    case DAFS_HIP_ENOMEM: return "dafs_hip: out of memory";
    case DAFS_HIP_ETOOLONG: return "dafs_hip: sequence too long for the device kernels";
    case DAFS_HIP_EOVERFLOW: return "dafs_hip: sparse output pool overflow";
    case DAFS_HIP_ELAUNCH: return "dafs_hip: kernel launch or execution failed";
    default: return "dafs_hip: unknown error";
  }
}

// wrapper.cpp:157-170: emission tables are filled for both cases of "ACGUTN"; every other byte
// keeps the defaults (emitPairs 1e-10, emitSingle 1e-5), which is class 6 here.
extern "C" uint8_t dafs_hip_residue_code(char c) {
  switch (c) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'U': case 'u': return 3;
    case 'T': case 't': return 4;
    case 'N': case 'n': return 5;
    default: return 6;
  }
}

// ProbCons RNA defaults (reference src/probconsRNA/Defaults.h:19-39) and the log tables the
// ProbabilisticModel constructor derives from them (ProbabilisticModel.h:55-88), with logf as
// there (LOG(float) resolves to std::log(float)).
extern "C" void dafs_hip_pairhmm3_default_model(dafs_pairhmm3_model* m) {
  static const float initDistrib[3] = {0.9588437676f, 0.0205782652f, 0.0205782652f};
  static const float gapOpen[2] = {0.0190259293f, 0.0190259293f};
  static const float gapExtend[2] = {0.3269913495f, 0.3269913495f};
  static const float emitSingle[6] = {0.2270790040f, 0.2422080040f, 0.2839320004f, 0.2464679927f, 0.2464679927f, 0.0003124650f};
  static const float emitPairs[6][6] = {
      {0.1487240046f, 0.0184142999f, 0.0361397006f, 0.0238473993f, 0.0238473993f, 0.0000375308f},
      {0.0184142999f, 0.1583919972f, 0.0275536999f, 0.0389291011f, 0.0389291011f, 0.0000815823f},
      {0.0361397006f, 0.0275536999f, 0.1979320049f, 0.0244289003f, 0.0244289003f, 0.0000824765f},
      {0.0238473993f, 0.0389291011f, 0.0244289003f, 0.1557479948f, 0.1557479948f, 0.0000743985f},
      {0.0238473993f, 0.0389291011f, 0.0244289003f, 0.1557479948f, 0.1557479948f, 0.0000743985f},
      {0.0000375308f, 0.0000815823f, 0.0000824765f, 0.0000743985f, 0.0000743985f, 0.0000263252f}};
  float trans[3][3] = {{0}};
  trans[0][0] = 1;
  trans[0][1] = gapOpen[0];
  trans[0][2] = gapOpen[1];
  trans[0][0] -= (gapOpen[0] + gapOpen[1]);
  trans[1][1] = gapExtend[0];
  trans[2][2] = gapExtend[1];
  trans[1][0] = 1 - gapExtend[0];
  trans[2][0] = 1 - gapExtend[1];
  for (int i = 0; i < 3; ++i) {
    m->init[i] = logf(initDistrib[i]);
    for (int j = 0; j < 3; ++j) m->trans[i][j] = logf(trans[i][j]);
  }
  const float pair_other = logf(1e-10f), single_other = logf(1e-5f);
  for (int i = 0; i < 7; ++i) {
    for (int j = 0; j < 8; ++j) m->match[i][j] = pair_other;
    for (int j = 0; j < 6 && i < 6; ++j) m->match[i][j] = logf(emitPairs[i][j]);
  }
  for (int i = 0; i < 8; ++i) m->ins[i] = i < 6 ? logf(emitSingle[i]) : single_other;
}

// ---------------------------------------------------------------------------------------------
extern "C" int dafs_hip_create(int device, dafs_hip_ctx** out) {
  if (!out) return DAFS_HIP_EINVAL;
  int n = 0;
  if (hip_check(hipGetDeviceCount(&n)) || n <= 0 || device < 0 || device >= n) return DAFS_HIP_ENODEV;
  if (hip_check(hipSetDevice(device))) return DAFS_HIP_ENODEV;
  dafs_hip_ctx* c = new dafs_hip_ctx();
  c->device = device;
  if (hip_check(hipStreamCreate(&c->stream))) { delete c; return DAFS_HIP_ENODEV; }
  *out = c;
  return DAFS_HIP_OK;
}

extern "C" void dafs_hip_destroy(dafs_hip_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  c->free_all();
  (void)hipStreamDestroy(c->stream);
  delete c;
}

extern "C" int dafs_hip_set_sequences(dafs_hip_ctx* c, uint32_t nseq, const char* const* seqs, const uint32_t* lens) {
  if (!c || !seqs || !lens || nseq == 0) return DAFS_HIP_EINVAL;
  if (hip_check(hipSetDevice(c->device))) return DAFS_HIP_ENODEV;
  c->len.assign(lens, lens + nseq);
  c->off.resize(nseq + 1);
  c->off[0] = 0;
  for (uint32_t i = 0; i < nseq; ++i) {
    if (lens[i] == 0) return DAFS_HIP_EINVAL;  // the reference reads seq[0] unconditionally (ProbabilisticModel.h:123-131)
    c->off[i + 1] = c->off[i] + lens[i];
  }
  c->seq.clear();
  c->seq.reserve(c->off[nseq]);
  std::vector<uint8_t> codes(c->off[nseq]);
  for (uint32_t i = 0; i < nseq; ++i) {
    c->seq.append(seqs[i], lens[i]);
    for (uint32_t k = 0; k < lens[i]; ++k) codes[c->off[i] + k] = dafs_hip_residue_code(seqs[i][k]);
  }
  int rc = c->codes.upload(codes.data(), codes.size(), c->stream);
  if (rc) return rc;
  c->align_valid = false;
  return hip_check(hipStreamSynchronize(c->stream)) ? DAFS_HIP_ELAUNCH : DAFS_HIP_OK;
}

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
  if (model != DAFS_ALIGN_PROBCONS) return DAFS_HIP_EINVAL;
  if (hip_check(hipSetDevice(c->device))) return DAFS_HIP_ENODEV;
  const uint32_t n = (uint32_t)c->len.size();
  const uint64_t all = (uint64_t)n * (n - 1) / 2;
  if (pair_end == 0) pair_end = all;
  if (pair_begin > pair_end || pair_end > all) return DAFS_HIP_EINVAL;
  const uint64_t np = pair_end - pair_begin;
  c->align_valid = false;
  c->pair_x.resize(np);
  c->pair_y.resize(np);
  if (np == 0) { c->align_valid = true; c->n_tasks = 0; return DAFS_HIP_OK; }
  {
    uint32_t i, j;
    pair_from_index(pair_begin, n, &i, &j);
    for (uint64_t p = 0; p < np; ++p) {
      c->pair_x[p] = i;
      c->pair_y[p] = j;
      if (++j == n) { ++i; j = i + 1; }
    }
  }
  // processing order: longest first (cost ~ len1*len2), so the work queue balances the tail
  std::vector<uint32_t> order(np);
  std::iota(order.begin(), order.end(), 0u);
  std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) {
    const uint64_t ca = (uint64_t)c->len[c->pair_x[a]] * c->len[c->pair_y[a]];
    const uint64_t cb = (uint64_t)c->len[c->pair_x[b]] * c->len[c->pair_y[b]];
    return ca > cb;
  });
  std::vector<dafs_pair_task> tasks(np);
  std::vector<uint64_t> rp_off(np);
  uint32_t max1 = 0, max2 = 0;
  uint64_t rp_total = 0, est = 0;
  // rp_off is laid out in shard order so dafs_hip_align_fetch can copy it out unchanged
  std::vector<uint64_t> rp_by_pair(np);
  for (uint64_t p = 0; p < np; ++p) {
    rp_by_pair[p] = rp_total;
    rp_total += (uint64_t)c->len[c->pair_x[p]] + 1 + c->len[c->pair_y[p]] + 1;
  }
  for (uint64_t k = 0; k < np; ++k) {
    const uint32_t p = order[k];
    const uint32_t x = c->pair_x[p], y = c->pair_y[p];
    tasks[k] = {c->off[x], c->len[x], c->off[y], c->len[y]};
    rp_off[k] = rp_by_pair[p];
    max1 = std::max(max1, c->len[x]);
    max2 = std::max(max2, c->len[y]);
    est += 2ull * std::min(c->len[x], c->len[y]) * 24;
  }
  dafs_pairhmm_plan plan;
  int rc = dafs_hipk_pairhmm_plan((uint32_t)np, max1, max2, &plan);
  if (rc) return rc;

  c->task_order = order;
  c->rp_by_pair = rp_by_pair;
  c->rp_total = rp_total;
  c->n_tasks = np;
  if ((rc = c->tasks.upload(tasks.data(), np, c->stream))) return rc;
  if ((rc = c->rp_off.upload(rp_off.data(), np, c->stream))) return rc;
  if ((rc = c->scratch.reserve(plan.scratch_bytes / sizeof(float)))) return rc;
  if ((rc = c->rowptr_pool.reserve(rp_total))) return rc;
  if ((rc = c->pair_off.reserve(np))) return rc;
  if ((rc = c->pair_nnz.reserve(np))) return rc;
  if ((rc = c->sim.reserve(np))) return rc;
  if ((rc = c->counters.reserve(4))) return rc;
  uint64_t cap = std::max<uint64_t>(est, 1024);
  if (c->pool_cap_hint > cap) cap = c->pool_cap_hint;

  for (int attempt = 0; attempt < 6; ++attempt) {
    if ((rc = c->ent_col.reserve(cap))) return rc;
    if ((rc = c->ent_val.reserve(cap))) return rc;
    if (hip_check(hipMemsetAsync(c->counters.ptr, 0, 4 * sizeof(unsigned long long), c->stream))) return DAFS_HIP_ELAUNCH;
    dafs_pairhmm3_args a;
    memset(&a, 0, sizeof a);
    a.codes = c->codes.ptr;
    a.tasks = c->tasks.ptr;
    a.ntasks = (uint32_t)np;
    a.th = th;
    a.scratch = c->scratch.ptr;
    a.queue = (uint32_t*)(c->counters.ptr + 1);
    a.rp_off = c->rp_off.ptr;
    a.rowptr_pool = c->rowptr_pool.ptr;
    a.ent_col = c->ent_col.ptr;
    a.ent_val = c->ent_val.ptr;
    a.pool_top = c->counters.ptr;
    a.pool_cap = cap;
    a.pair_off = c->pair_off.ptr;
    a.pair_nnz = c->pair_nnz.ptr;
    a.sim = c->sim.ptr;
    a.status = (int*)(c->counters.ptr + 2);
    dafs_hip_pairhmm3_default_model(&a.model);
    if ((rc = dafs_hipk_pairhmm3_launch(&a, &plan, c->stream))) return rc;
    unsigned long long host_cnt[4];
    if (hip_check(hipMemcpyAsync(host_cnt, c->counters.ptr, sizeof host_cnt, hipMemcpyDeviceToHost, c->stream))) return DAFS_HIP_ELAUNCH;
    if (hip_check(hipStreamSynchronize(c->stream))) return DAFS_HIP_ELAUNCH;
    const int status = (int)(host_cnt[2] & 0xffffffffu);
    if (status == 0) {
      c->pool_used = host_cnt[0];
      c->pool_cap_hint = cap;
      c->align_valid = true;
      c->plan = plan;
      return DAFS_HIP_OK;
    }
    if (status != DAFS_HIP_EOVERFLOW) return status;
    cap = std::max<uint64_t>(host_cnt[0], cap * 2);  // pool_top kept counting: exact requirement
  }
  return DAFS_HIP_EOVERFLOW;
}

extern "C" int dafs_hip_align_result_size(dafs_hip_ctx* c, uint64_t* npairs, uint64_t* total_nnz, uint64_t* total_rowptr) {
  if (!c || !c->align_valid) return DAFS_HIP_EINVAL;
  if (npairs) *npairs = c->n_tasks;
  if (total_nnz) *total_nnz = c->pool_used / 2;
  if (total_rowptr) *total_rowptr = c->rp_total;
  return DAFS_HIP_OK;
}

extern "C" int dafs_hip_align_fetch(dafs_hip_ctx* c, uint32_t* pair_x, uint32_t* pair_y, float* sim, uint32_t* nnz,
                                    uint32_t* rowptr, uint32_t* col, float* val) {
  if (!c || !c->align_valid) return DAFS_HIP_EINVAL;
  if (hip_check(hipSetDevice(c->device))) return DAFS_HIP_ENODEV;
  const uint64_t np = c->n_tasks;
  if (np == 0) return DAFS_HIP_OK;
  if (pair_x) memcpy(pair_x, c->pair_x.data(), np * sizeof(uint32_t));
  if (pair_y) memcpy(pair_y, c->pair_y.data(), np * sizeof(uint32_t));
  std::vector<float> h_sim(np);
  std::vector<uint32_t> h_nnz(np);
  std::vector<uint64_t> h_off(np);
  if (hip_check(hipMemcpy(h_sim.data(), c->sim.ptr, np * sizeof(float), hipMemcpyDeviceToHost))) return DAFS_HIP_ELAUNCH;
  if (hip_check(hipMemcpy(h_nnz.data(), c->pair_nnz.ptr, np * sizeof(uint32_t), hipMemcpyDeviceToHost))) return DAFS_HIP_ELAUNCH;
  if (hip_check(hipMemcpy(h_off.data(), c->pair_off.ptr, np * sizeof(uint64_t), hipMemcpyDeviceToHost))) return DAFS_HIP_ELAUNCH;
  // device arrays are in task (processing) order; outputs are in shard order
  std::vector<uint64_t> task_of_pair(np);
  for (uint64_t k = 0; k < np; ++k) task_of_pair[c->task_order[k]] = k;
  if (sim) for (uint64_t p = 0; p < np; ++p) sim[p] = h_sim[task_of_pair[p]];
  if (nnz) for (uint64_t p = 0; p < np; ++p) nnz[p] = h_nnz[task_of_pair[p]];
  if (rowptr && hip_check(hipMemcpy(rowptr, c->rowptr_pool.ptr, c->rp_total * sizeof(uint32_t), hipMemcpyDeviceToHost))) return DAFS_HIP_ELAUNCH;
  if (col || val) {
    std::vector<uint32_t> h_col;
    std::vector<float> h_val;
    if (col) { h_col.resize(c->pool_used); if (hip_check(hipMemcpy(h_col.data(), c->ent_col.ptr, c->pool_used * sizeof(uint32_t), hipMemcpyDeviceToHost))) return DAFS_HIP_ELAUNCH; }
    if (val) { h_val.resize(c->pool_used); if (hip_check(hipMemcpy(h_val.data(), c->ent_val.ptr, c->pool_used * sizeof(float), hipMemcpyDeviceToHost))) return DAFS_HIP_ELAUNCH; }
    uint64_t w = 0;
    for (uint64_t p = 0; p < np; ++p) {
      const uint64_t k = task_of_pair[p];
      const uint64_t n2 = 2ull * h_nnz[k];
      if (col) memcpy(col + w, h_col.data() + h_off[k], n2 * sizeof(uint32_t));
      if (val) memcpy(val + w, h_val.data() + h_off[k], n2 * sizeof(float));
      w += n2;
    }
  }
  return DAFS_HIP_OK;
}
