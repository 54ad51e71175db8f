// dafs_amd/csrc/capi_dev.cpp -- the sparse stores of a context exported to / installed from DEVICE buffers
// (include/dafs_hip.h, "device-resident exchange"): what a multi-GPU run puts between its phases instead of host
// round trips.  One process per GPU shards DAFS::run's phase 1 (reference src/dafs.cpp:1787-1827: the N folds of
// fold.cpp:66-67, the N(N-1)/2 pair jobs of align.cpp:46-50, the output pairs of relax_matching_probability,
// dafs.cpp:265-315), and the shards travel by all-gather over RCCL straight from and into these buffers.
#include <hip/hip_runtime.h>
#include <string.h>
#include <algorithm>
#include <vector>

#include "../../include/dafs_hip.h"
#include "ctx.h"
#include "hip_util.h"
#include "store_dev.h"

using namespace dafs;

namespace {
int d2d(void* dst, const void* src, size_t bytes, hipStream_t st) {
  if (!bytes) return DAFS_HIP_OK;
  return hip_check(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, st)) ? DAFS_HIP_ELAUNCH : DAFS_HIP_OK;
}
}  // namespace

extern "C" int dafs_hip_mp_export_dev(dafs_hip_ctx* c, int relaxed, uint64_t first, uint64_t count, uint32_t* nnz, uint32_t* rowptr, uint32_t* col, float* val,
                                      float* sim, uint64_t cap_entries, uint64_t* n_rowptr, uint64_t* n_entries) {
  if (!c || relaxed < 0 || relaxed > 1 || !c->mp[relaxed].valid || !n_rowptr || !n_entries) return DAFS_HIP_EINVAL;
  if (hip_check(hipSetDevice(c->device))) return DAFS_HIP_ENODEV;
  const mp_store& st = c->mp[relaxed];
  if (first > st.n_tasks || count > st.n_tasks - first) return DAFS_HIP_EINVAL;
  *n_rowptr = 0; *n_entries = 0;
  if (count == 0) return DAFS_HIP_OK;
  if (!nnz || !rowptr || (sim && relaxed != 0)) return DAFS_HIP_EINVAL;
  const uint64_t rp0 = st.rp_by_pair[first], rp1 = first + count < st.n_tasks ? st.rp_by_pair[first + count] : st.rp_total;
  int rc;
  if ((rc = gather_tasks_launch(st.d_task_of_pair.ptr, first, count, st.pair_nnz.ptr, sim ? c->task_sim.ptr : nullptr, nnz, sim, c->stream))) return rc;
  if ((rc = c->work2.reserve((count + 1) * sizeof(uint64_t) + 64))) return rc;
  uint64_t* prefix = (uint64_t*)c->work2.ptr;
  if ((rc = scan_excl_launch(nnz, 2u, prefix, count, c->stream))) return rc;
  uint64_t total = 0;
  if (hip_check(hipMemcpyAsync(&total, prefix + count, sizeof total, hipMemcpyDeviceToHost, c->stream)) || hip_check(hipStreamSynchronize(c->stream))) return DAFS_HIP_ELAUNCH;
  if (total > cap_entries) return DAFS_HIP_EOVERFLOW;
  if (total && (!col || !val)) return DAFS_HIP_EINVAL;
  if ((rc = d2d(rowptr, st.rowptr_pool.ptr + rp0, (rp1 - rp0) * sizeof(uint32_t), c->stream))) return rc;
  if (total && (rc = mp_pack_launch(st.d_task_of_pair.ptr, first, count, st.pair_off.ptr, st.pair_nnz.ptr, st.col.ptr, st.val.ptr, prefix, col, val, c->stream))) return rc;
  if (hip_check(hipStreamSynchronize(c->stream))) return DAFS_HIP_ELAUNCH;  // the caller's streams may read the buffers now
  *n_rowptr = rp1 - rp0;
  *n_entries = total;
  return DAFS_HIP_OK;
}

extern "C" int dafs_hip_mp_install_dev(dafs_hip_ctx* c, int relaxed, const uint32_t* nnz, const uint32_t* rowptr, const uint32_t* col, const float* val,
                                       const float* sim, uint64_t n_entries) {
  if (!c || c->len.size() < 2 || relaxed < 0 || relaxed > 1 || !nnz || !rowptr) return DAFS_HIP_EINVAL;
  if (relaxed == 0 && !sim) return DAFS_HIP_EINVAL;
  if (relaxed == 1 && (!c->mp[0].valid || c->sim.empty())) return DAFS_HIP_EINVAL;
  if (n_entries && (!col || !val)) return DAFS_HIP_EINVAL;
  if (hip_check(hipSetDevice(c->device))) return DAFS_HIP_ENODEV;
  const uint32_t n = (uint32_t)c->len.size();
  const uint64_t np = (uint64_t)n * (n - 1) / 2;
  mp_store& st = c->mp[relaxed];
  st.valid = false;
  if (relaxed == 0) { c->mp[1].valid = false; c->cur_mp = 0; c->sim.clear(); }
  st.pair_x.resize(np); st.pair_y.resize(np); st.task_of_pair.resize(np); st.rp_by_pair.resize(np);
  st.n_tasks = np;
  uint64_t rp_total = 0, p = 0;
  for (uint32_t x = 0; x < n; ++x)
    for (uint32_t y = x + 1; y < n; ++y, ++p) {
      st.pair_x[p] = x; st.pair_y[p] = y; st.task_of_pair[p] = (uint32_t)p;
      st.rp_by_pair[p] = rp_total;
      rp_total += (uint64_t)c->len[x] + 1 + c->len[y] + 1;
    }
  int rc;
  st.rp_total = rp_total;
  st.pool_used = n_entries;
  st.pool_cap_hint = std::max<uint64_t>(st.pool_cap_hint, n_entries);
  if ((rc = st.rowptr_pool.reserve(rp_total))) return rc;
  if ((rc = st.col.reserve(n_entries + 1))) return rc;
  if ((rc = st.val.reserve(n_entries + 1))) return rc;
  if ((rc = st.pair_nnz.reserve(np))) return rc;
  if ((rc = st.pair_off.reserve(np + 1))) return rc;
  if ((rc = d2d(st.rowptr_pool.ptr, rowptr, rp_total * sizeof(uint32_t), c->stream))) return rc;
  if ((rc = d2d(st.col.ptr, col, n_entries * sizeof(uint32_t), c->stream))) return rc;
  if ((rc = d2d(st.val.ptr, val, n_entries * sizeof(float), c->stream))) return rc;
  if ((rc = d2d(st.pair_nnz.ptr, nnz, np * sizeof(uint32_t), c->stream))) return rc;
  if ((rc = scan_excl_launch(st.pair_nnz.ptr, 2u, st.pair_off.ptr, np, c->stream))) return rc;  // entries of pair p start at twice the sum of nnz before it
  {  // the entry count the caller states must be what the counts add up to
    uint64_t total = 0;
    if (hip_check(hipMemcpyAsync(&total, st.pair_off.ptr + np, sizeof total, hipMemcpyDeviceToHost, c->stream)) || hip_check(hipStreamSynchronize(c->stream))) return DAFS_HIP_ELAUNCH;
    if (total != n_entries) return DAFS_HIP_EINVAL;
  }
  if ((rc = st.rp_off.upload(st.rp_by_pair.data(), np, c->stream))) return rc;
  if ((rc = st.d_task_of_pair.upload(st.task_of_pair.data(), np, c->stream))) return rc;
  if ((rc = c->d_pair_x.upload(st.pair_x.data(), np, c->stream))) return rc;
  if ((rc = c->d_pair_y.upload(st.pair_y.data(), np, c->stream))) return rc;
  if (relaxed == 0) {
    if ((rc = c->task_sim.reserve(np))) return rc;
    if ((rc = d2d(c->task_sim.ptr, sim, np * sizeof(float), c->stream))) return rc;
    if ((rc = c->d_sim.reserve((size_t)n * n))) return rc;
    if (hip_check(hipMemsetAsync(c->d_sim.ptr, 0, (size_t)n * n * sizeof(float), c->stream))) return DAFS_HIP_ELAUNCH;
    if ((rc = sim_matrix_launch(c->d_pair_x.ptr, c->d_pair_y.ptr, c->task_sim.ptr, np, n, c->d_sim.ptr, c->stream))) return rc;
    c->sim.assign((size_t)n * n, 0.0f);  // the guide tree is host work (dafs_host_build_tree): N * N floats come down
    if (hip_check(hipMemcpyAsync(c->sim.data(), c->d_sim.ptr, c->sim.size() * sizeof(float), hipMemcpyDeviceToHost, c->stream)) ||
        hip_check(hipStreamSynchronize(c->stream))) return DAFS_HIP_ELAUNCH;
  } else {
    if (hip_check(hipStreamSynchronize(c->stream))) return DAFS_HIP_ELAUNCH;  // the caller may reuse its buffers
    c->cur_mp = 1;
  }
  st.valid = true;
  return DAFS_HIP_OK;
}

extern "C" int dafs_hip_bp_export_dev(dafs_hip_ctx* c, uint32_t* rowptr, uint32_t* col, float* val, uint64_t cap_entries, uint64_t* n_rowptr, uint64_t* n_entries) {
  if (!c || c->len.empty() || !c->bp[0].valid || !rowptr || !n_rowptr || !n_entries) return DAFS_HIP_EINVAL;
  if (hip_check(hipSetDevice(c->device))) return DAFS_HIP_ENODEV;
  const bp_store& st = c->bp[0];
  const uint32_t n = (uint32_t)c->len.size();
  if (st.total_nnz > cap_entries) return DAFS_HIP_EOVERFLOW;
  if (st.total_nnz && (!col || !val)) return DAFS_HIP_EINVAL;
  int rc;
  if ((rc = d2d(rowptr, st.rowptr.ptr, c->seq_rp_off[n] * sizeof(uint32_t), c->stream))) return rc;
  if ((rc = c->work2.reserve(((size_t)n + 1) * sizeof(uint64_t) + 64))) return rc;
  uint64_t* prefix = (uint64_t*)c->work2.ptr;
  if ((rc = scan_excl_launch(st.nnz.ptr, 1u, prefix, n, c->stream))) return rc;
  if ((rc = bp_pack_launch(n, st.bp_off.ptr, st.nnz.ptr, st.col.ptr, st.val.ptr, prefix, col, val, c->stream))) return rc;
  if (hip_check(hipStreamSynchronize(c->stream))) return DAFS_HIP_ELAUNCH;
  *n_rowptr = c->seq_rp_off[n];
  *n_entries = st.total_nnz;
  return DAFS_HIP_OK;
}

extern "C" int dafs_hip_set_bp_dev(dafs_hip_ctx* c, uint32_t nblocks, const uint32_t* seq_of_block, const uint32_t* rowptr, const uint32_t* col, const float* val,
                                   uint64_t n_entries) {
  if (!c || c->len.empty() || !seq_of_block || !rowptr || nblocks != c->len.size()) return DAFS_HIP_EINVAL;
  if (n_entries && (!col || !val)) return DAFS_HIP_EINVAL;
  if (hip_check(hipSetDevice(c->device))) return DAFS_HIP_ENODEV;
  const uint32_t n = nblocks;
  bp_store& st = c->bp[0];
  st.valid = false;
  c->bp[1].valid = false;
  c->cur_bp = 0;
  // where each block's row pointers start in the gathered array, and which sequence it is
  std::vector<uint64_t> blk_rp(n);
  std::vector<uint32_t> blk_len(n);
  std::vector<uint8_t> seen(n, 0);
  uint64_t rp = 0;
  for (uint32_t k = 0; k < n; ++k) {
    const uint32_t x = seq_of_block[k];
    if (x >= n || seen[x]) return DAFS_HIP_EINVAL;
    seen[x] = 1;
    blk_rp[k] = rp; blk_len[k] = c->len[x];
    rp += (uint64_t)c->len[x] + 1;
  }
  if (rp != c->seq_rp_off[n]) return DAFS_HIP_EINVAL;
  int rc;
  if ((rc = st.rowptr.reserve(rp))) return rc;
  if ((rc = st.col.reserve(n_entries + 1))) return rc;
  if ((rc = st.val.reserve(n_entries + 1))) return rc;
  if ((rc = st.nnz.reserve(n))) return rc;
  if ((rc = st.bp_off.reserve(n + 1))) return rc;
  // row pointers: every block to its sequence's place (the store keeps them in sequence order, dafs_hip_bp_fetch copies them
  // out as they lie); entries stay in block order, each sequence knows where its own start (bp_off)
  for (uint32_t k = 0; k < n; ++k) {
    const uint32_t x = seq_of_block[k];
    if ((rc = d2d(st.rowptr.ptr + c->seq_rp_off[x], rowptr + blk_rp[k], ((size_t)c->len[x] + 1) * sizeof(uint32_t), c->stream))) return rc;
  }
  if ((rc = d2d(st.col.ptr, col, n_entries * sizeof(uint32_t), c->stream))) return rc;
  if ((rc = d2d(st.val.ptr, val, n_entries * sizeof(float), c->stream))) return rc;
  if ((rc = st.rp_off.upload(c->seq_rp_off.data(), n + 1, c->stream))) return rc;
  // per block: entries (its last row pointer) and first entry (blocks follow one another in the gathered pools)
  const size_t words = (size_t)n * 8 + 64;
  if ((rc = c->work.reserve(words * 4))) return rc;
  uint8_t* w = c->work.ptr;
  uint64_t* d_blk_rp = (uint64_t*)w; w += (size_t)n * 8;
  uint64_t* d_off_by_blk = (uint64_t*)w; w += ((size_t)n + 1) * 8;
  uint32_t* d_blk_len = (uint32_t*)w; w += (size_t)n * 4;
  uint32_t* d_seq_of_blk = (uint32_t*)w; w += (size_t)n * 4;
  uint32_t* d_nnz_by_blk = (uint32_t*)w;
  if (hip_check(hipMemcpyAsync(d_blk_rp, blk_rp.data(), (size_t)n * 8, hipMemcpyHostToDevice, c->stream))) return DAFS_HIP_ELAUNCH;
  if (hip_check(hipMemcpyAsync(d_blk_len, blk_len.data(), (size_t)n * 4, hipMemcpyHostToDevice, c->stream))) return DAFS_HIP_ELAUNCH;
  if (hip_check(hipMemcpyAsync(d_seq_of_blk, seq_of_block, (size_t)n * 4, hipMemcpyHostToDevice, c->stream))) return DAFS_HIP_ELAUNCH;
  if ((rc = bp_block_nnz_launch(rowptr, d_blk_rp, d_blk_len, n, d_nnz_by_blk, c->stream))) return rc;
  if ((rc = scan_excl_launch(d_nnz_by_blk, 1u, d_off_by_blk, n, c->stream))) return rc;
  if ((rc = bp_by_seq_launch(d_seq_of_blk, n, d_nnz_by_blk, d_off_by_blk, st.nnz.ptr, st.bp_off.ptr, c->stream))) return rc;
  uint64_t total = 0;
  if (hip_check(hipMemcpyAsync(&total, d_off_by_blk + n, sizeof total, hipMemcpyDeviceToHost, c->stream)) || hip_check(hipStreamSynchronize(c->stream))) return DAFS_HIP_ELAUNCH;
  if (total != n_entries) return DAFS_HIP_EINVAL;
  st.total_nnz = total;
  st.valid = true;
  return DAFS_HIP_OK;
}
