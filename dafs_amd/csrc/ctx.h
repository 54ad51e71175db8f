// dafs_amd/csrc/ctx.h -- the context object behind the L1 entry points: device buffers that stay
// resident between calls (sequence codes, the sparse posterior pools, similarity scores).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>

#include "../../include/dafs_hip.h"
#include "hip_util.h"

namespace dafs {

// grow-only device array
template <class T>
struct dev_buf {
  T* ptr = nullptr;
  size_t cap = 0;
  int reserve(size_t n) {
    if (n <= cap) return DAFS_HIP_OK;
    if (ptr) (void)hipFree(ptr);
    ptr = nullptr;
    cap = 0;
    size_t want = n + n / 8 + 64;
    if (hip_check(hipMalloc((void**)&ptr, want * sizeof(T)))) return DAFS_HIP_ENOMEM;
    cap = want;
    return DAFS_HIP_OK;
  }
  int upload(const T* host, size_t n, hipStream_t st) {
    int rc = reserve(n);
    if (rc) return rc;
    if (n && hip_check(hipMemcpyAsync(ptr, host, n * sizeof(T), hipMemcpyHostToDevice, st))) return DAFS_HIP_ELAUNCH;
    if (hip_check(hipStreamSynchronize(st))) return DAFS_HIP_ELAUNCH;  // host vector may die after return
    return DAFS_HIP_OK;
  }
  void release() {
    if (ptr) (void)hipFree(ptr);
    ptr = nullptr;
    cap = 0;
  }
};

}  // namespace dafs

struct dafs_hip_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  // sequences
  std::string seq;
  std::vector<uint32_t> len, off;
  dafs::dev_buf<uint8_t> codes;
  // alignment-posterior shard
  bool align_valid = false;
  uint64_t n_tasks = 0, rp_total = 0, pool_used = 0, pool_cap_hint = 0;
  std::vector<uint32_t> pair_x, pair_y, task_order;
  std::vector<uint64_t> rp_by_pair;
  dafs_pairhmm_plan plan{};
  dafs::dev_buf<dafs_pair_task> tasks;
  dafs::dev_buf<uint64_t> rp_off, pair_off;
  dafs::dev_buf<float> scratch, ent_val, sim;
  dafs::dev_buf<uint32_t> rowptr_pool, ent_col, pair_nnz;
  dafs::dev_buf<unsigned long long> counters;

  void free_all() {
    codes.release(); tasks.release(); rp_off.release(); pair_off.release(); scratch.release();
    ent_val.release(); sim.release(); rowptr_pool.release(); ent_col.release(); pair_nnz.release();
    counters.release();
  }
};
