// dafs_amd/csrc/ctx.h -- the context object behind the L1 entry points: device buffers that stay
// resident between calls (sequence codes, the sparse posterior stores, similarity scores).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>

#include "../../include/dafs_hip.h"
#include "hip_util.h"
#include "sparse_view.h"
#include "dd.h"

namespace dafs {

// grow-only device array
template <class T>
struct dev_buf {
  T* ptr = nullptr;
  size_t cap = 0;
  int reserve(size_t n) {
    if (n <= cap && ptr) return DAFS_HIP_OK;
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
  int download(T* host, size_t n) const {
    if (n && hip_check(hipMemcpy(host, ptr, n * sizeof(T), hipMemcpyDeviceToHost))) return DAFS_HIP_ELAUNCH;
    return DAFS_HIP_OK;
  }
  void release() {
    if (ptr) (void)hipFree(ptr);
    ptr = nullptr;
    cap = 0;
  }
};

// Matching-probability store: per task the rows of mp[x][y] followed by those of mp[y][x].
struct mp_store {
  bool valid = false;
  uint64_t n_tasks = 0, rp_total = 0, pool_used = 0, pool_cap_hint = 0;
  std::vector<uint32_t> pair_x, pair_y;   // per pair of the shard, shard order
  std::vector<uint32_t> task_of_pair;     // shard-order pair -> task (processing order)
  std::vector<uint64_t> rp_by_pair;
  dev_buf<uint32_t> rowptr_pool, col, pair_nnz, d_task_of_pair;
  dev_buf<float> val;
  dev_buf<uint64_t> pair_off, rp_off;
  mp_store_dev view(const uint32_t* d_len, uint32_t nseq) const {
    mp_store_dev v;
    v.rowptr_pool = rowptr_pool.ptr; v.col = col.ptr; v.val = val.ptr; v.pair_off = pair_off.ptr;
    v.pair_nnz = pair_nnz.ptr; v.rp_off = rp_off.ptr; v.task_of_pair = d_task_of_pair.ptr; v.len = d_len; v.nseq = nseq;
    return v;
  }
  void release() {
    rowptr_pool.release(); col.release(); pair_nnz.release(); d_task_of_pair.release(); val.release();
    pair_off.release(); rp_off.release(); valid = false;
  }
};

// Base-pairing store: per sequence x the rows i -> (j > i, p).
struct bp_store {
  bool valid = false;
  uint64_t total_nnz = 0;
  dev_buf<uint32_t> rowptr, col, nnz;
  dev_buf<float> val;
  dev_buf<uint64_t> rp_off, bp_off;
  bp_store_dev view() const {
    bp_store_dev v;
    v.rowptr = rowptr.ptr; v.col = col.ptr; v.val = val.ptr; v.rp_off = rp_off.ptr; v.bp_off = bp_off.ptr;
    return v;
  }
  void release() { rowptr.release(); col.release(); nnz.release(); val.release(); rp_off.release(); bp_off.release(); valid = false; }
};

}  // namespace dafs

struct dafs_hip_ctx {
  int device = 0;
  int num_cus = 256;  // compute units of the device (co-residency bound of the split node solver)
  hipStream_t stream = nullptr;
  hipStream_t fold_stream = nullptr;  // the per-sequence folding runs here, beside the pair / consistency kernels
  bool fold_pending = false;          // dafs_hip_fold_posteriors_begin without its _end
  float fold_th = 0.0f;
  std::vector<uint8_t> fold_batch;    // the cf_batch of the pending job (contrafold.h), kept opaque here
  // sequences
  std::string seq;
  std::vector<uint32_t> len, off;
  std::vector<uint64_t> seq_rp_off;  // per sequence: first row pointer in a per-sequence CSR (sum of len+1)
  dafs::dev_buf<uint8_t> codes;
  dafs::dev_buf<uint32_t> d_len;
  dafs::dev_buf<uint64_t> d_seq_rp_off;
  // pair-HMM launch workspace
  dafs_pairhmm_plan plan{};
  dafs::dev_buf<dafs_pair_task> tasks;
  dafs::dev_buf<float> scratch, task_sim;
  dafs::dev_buf<unsigned long long> counters;
  // stores: [0] as computed by the models, [1] after the consistency transforms
  dafs::mp_store mp[2];
  dafs::bp_store bp[2];
  int cur_mp = 0, cur_bp = 0;
  // similarity matrix (host copy + device dense N*N)
  std::vector<float> sim;
  dafs::dev_buf<float> d_sim;
  dafs::dev_buf<uint32_t> d_pair_x, d_pair_y;
  // CONTRAfold workspaces
  bool cf_params_ready = false;
  dafs::dev_buf<uint8_t> d_cf_params, cf_seqs, cf_codes;
  dafs::dev_buf<int> cf_iws, cf_cons;
  dafs::dev_buf<float> cf_fws, cf_post, cf_logz;
  dafs::dev_buf<unsigned long long> cf_stamps;
  // progressive phase workspaces
  dafs::dev_buf<uint8_t> work, work2;
  dafs::dev_buf<dafs::dd_node> d_nodes;
  dafs::dev_buf<uint32_t> d_paused;  // per node of a launch: still unfinished
  // resident tree nodes (dafs_hip_nodes_open / _advance / _result / _close): device memory that lives until
  // _close, in large chunks that are kept for the next phase
  struct dd_chunk { uint8_t* ptr; size_t cap, used; };
  std::vector<dd_chunk> dd_chunks;
  struct dd_open_node { dafs::dd_node nd; size_t lds, split_lds; bool finished; };
  std::vector<dd_open_node> dd_open;
  uint8_t* dd_alloc(size_t bytes) {
    bytes = (bytes + 255) & ~(size_t)255;
    for (dd_chunk& ch : dd_chunks)
      if (ch.cap - ch.used >= bytes) { uint8_t* p = ch.ptr + ch.used; ch.used += bytes; return p; }
    dd_chunk ch;
    ch.cap = bytes > ((size_t)256 << 20) ? bytes : ((size_t)256 << 20);
    ch.used = bytes;
    if (dafs::hip_check(hipMalloc((void**)&ch.ptr, ch.cap))) return nullptr;
    dd_chunks.push_back(ch);
    return ch.ptr;
  }
  void dd_reset() { for (dd_chunk& ch : dd_chunks) ch.used = 0; dd_open.clear(); }
  void dd_release() { for (dd_chunk& ch : dd_chunks) (void)hipFree(ch.ptr); dd_chunks.clear(); dd_open.clear(); }
  uint32_t max_len() const { uint32_t m = 0; for (uint32_t l : len) m = l > m ? l : m; return m; }

  void free_all() {
    codes.release(); d_len.release(); d_seq_rp_off.release(); tasks.release(); scratch.release(); task_sim.release();
    counters.release(); d_sim.release(); d_pair_x.release(); d_pair_y.release(); work.release(); work2.release(); d_nodes.release(); d_paused.release(); dd_release();
    d_cf_params.release(); cf_seqs.release(); cf_codes.release(); cf_iws.release(); cf_cons.release(); cf_fws.release(); cf_post.release(); cf_logz.release();
    for (int k = 0; k < 2; ++k) { mp[k].release(); bp[k].release(); }
  }
};
