// dafs_amd/csrc/ctx.h -- the context object behind the L1 entry points: device buffers that stay
// resident between calls (sequence codes, the sparse posterior stores, similarity scores).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <iterator>
#include <map>
#include <string>
#include <vector>

#include "../../include/dafs_hip.h"
#include "hip_util.h"
#include "sparse_view.h"
#include "dd.h"
#include "stage.h"

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
    v.rowptr_pool = rowptr_pool.ptr; v.col = col.ptr; v.val = val.ptr; v.ent2 = nullptr; v.ident2 = nullptr; v.ident_rp = nullptr; v.pair_off = pair_off.ptr;
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
  hipStream_t node_stream = nullptr;  // dafs_hip_nodes_round: new nodes are set up and started here while the open ones advance on `stream`
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
  dafs::bp_store bp_rows;  // dafs_hip_update_basepairing: constrained posteriors of the rows of one alignment (store index = row)
  int cur_mp = 0, cur_bp = 0;
  // similarity matrix (host copy + device dense N*N)
  std::vector<float> sim;
  dafs::dev_buf<float> d_sim;
  dafs::dev_buf<uint32_t> d_pair_x, d_pair_y;
  dafs::dev_buf<uint2> mp_ident2;  // rows of an identity matrix, {k, 1.0f}
  dafs::dev_buf<uint2> pct_tasks;  // workgroup order of k_pct_rows (pct_task_order)
  dafs::dev_buf<uint2> mp_ent2;  // interleaved copy of the un-relaxed matching store's entries (consistency transforms)
  // CONTRAfold workspaces
  bool cf_params_ready = false;
  dafs::dev_buf<uint8_t> d_cf_params, cf_seqs, cf_codes;
  dafs::dev_buf<int> cf_iws, cf_cons;
  dafs::dev_buf<float> cf_fws, cf_post, cf_logz;
  dafs::dev_buf<unsigned long long> cf_stamps;
  dafs::stage_recorder stages;  // dafs_hip_stage_timing / _report
  // progressive phase workspaces
  dafs::dev_buf<uint8_t> work, work2;
  dafs::dev_buf<dafs::dd_node> d_nodes;
  dafs::dev_buf<uint32_t> d_paused;  // per node of a launch: still unfinished
  dafs::dev_buf<dafs::dd_node> d_nodes2;  // the same pair for the second lane of dafs_hip_nodes_round
  dafs::dev_buf<uint32_t> d_paused2;
  dafs::dev_buf<uint32_t> d_pack_off[2], d_pack[2];  // per lane: where each node's result words go in the packed buffer, and that buffer
  dafs::dev_buf<unsigned long long> d_tref;  // start tick of a round's first launch: the lanes of a round share one deadline
  // pinned landing places of the per-node words of a launch, one per lane: a copy into pageable memory would make the
  // "asynchronous" copy wait for the kernel in front of it, and the second lane could not start beside the first
  uint32_t* h_paused[2] = {nullptr, nullptr};
  size_t h_paused_cap[2] = {0, 0};
  uint32_t* pinned_words(int lane, size_t n) {
    if (n <= h_paused_cap[lane] && h_paused[lane]) return h_paused[lane];
    if (h_paused[lane]) (void)hipHostFree(h_paused[lane]);
    h_paused[lane] = nullptr; h_paused_cap[lane] = 0;
    const size_t want = n + n / 2 + 256;
    if (dafs::hip_check(hipHostMalloc((void**)&h_paused[lane], want * sizeof(uint32_t), hipHostMallocDefault))) return nullptr;
    h_paused_cap[lane] = want;
    return h_paused[lane];
  }
  // resident tree nodes (dafs_hip_nodes_open / _advance / _result / _close).  Their device memory comes from large
  // chunks that are kept for the next phase; a node's blocks go back to an address-ordered free list (neighbours
  // merged) as soon as its result has been copied out (dafs_hip_nodes_result), so a progressive run holds the nodes
  // that are open, not every node of the tree (a node is ~40 L^2 bytes: 3 GB at 8600 columns).
  struct dd_chunk { uint8_t* ptr; size_t cap; };
  std::vector<dd_chunk> dd_chunks;
  std::map<uint8_t*, size_t> dd_free_blocks;  // start -> bytes
  size_t dd_in_use = 0, dd_peak = 0;
  struct dd_open_node {
    dafs::dd_node nd;
    size_t lds = 0, split_lds = 0;
    bool finished = false;
    uint8_t* blk[2] = {nullptr, nullptr};
    size_t blk_bytes[2] = {0, 0};
    bool released = false;   // its blocks went back to the free list (dafs_hip_nodes_result)
    bool no_split = false;   // a launch lost this node's folding workgroups once (k_dd_solve): keep it on one workgroup
    bool in_flight = false;  // part of a launch that has not been collected yet: its blocks must not be freed
    std::vector<uint32_t> result;  // the node's result words, brought along by the launch it finished in
  };
  std::vector<dd_open_node> dd_open;
  uint32_t dd_wgs_in_flight[2] = {0, 0};  // per lane: workgroups of the uncollected launch (split launches of both lanes must fit the device together)
  uint32_t dd_demotions = 0;              // nodes whose folders were lost and that went on in the one-workgroup form (since nodes_close)
  // true when [p, p + bytes) overlaps a block of an open node that has not been released (a free of such a range is a bug)
  bool dd_range_live(const uint8_t* p, size_t bytes, const dd_open_node* except) const {
    for (const dd_open_node& on : dd_open) {
      if (&on == except || on.released) continue;
      for (int k = 0; k < 2; ++k)
        if (on.blk[k] && p < on.blk[k] + on.blk_bytes[k] && on.blk[k] < p + bytes) return true;
    }
    return false;
  }
  void dd_free(uint8_t* p, size_t bytes) {
    if (!p || !bytes) return;
    bytes = (bytes + 255) & ~(size_t)255;
    dd_in_use -= bytes;
    auto it = dd_free_blocks.emplace(p, bytes).first;
    auto nx = std::next(it);
    if (nx != dd_free_blocks.end() && it->first + it->second == nx->first && same_chunk(it->first, nx->first)) { it->second += nx->second; dd_free_blocks.erase(nx); }
    if (it != dd_free_blocks.begin()) {
      auto pv = std::prev(it);
      if (pv->first + pv->second == it->first && same_chunk(pv->first, it->first)) { pv->second += it->second; dd_free_blocks.erase(it); }
    }
  }
  bool same_chunk(const uint8_t* a, const uint8_t* b) const {
    for (const dd_chunk& ch : dd_chunks)
      if (a >= ch.ptr && a < ch.ptr + ch.cap) return b >= ch.ptr && b < ch.ptr + ch.cap;
    return false;
  }
  uint8_t* dd_alloc(size_t bytes) {
    bytes = (bytes + 255) & ~(size_t)255;
    auto best = dd_free_blocks.end();
    for (auto it = dd_free_blocks.begin(); it != dd_free_blocks.end(); ++it)
      if (it->second >= bytes && (best == dd_free_blocks.end() || it->second < best->second)) best = it;
    if (best == dd_free_blocks.end()) {
      // nothing fits: chunks that are entirely free go back to the device first (a wide node may have left a chunk of
      // several GB behind that no later node fills)
      for (size_t k = 0; k < dd_chunks.size();) {
        auto it = dd_free_blocks.find(dd_chunks[k].ptr);
        if (it != dd_free_blocks.end() && it->second == dd_chunks[k].cap) {
          dd_free_blocks.erase(it);
          (void)hipFree(dd_chunks[k].ptr);
          dd_chunks.erase(dd_chunks.begin() + (long)k);
        } else {
          ++k;
        }
      }
      dd_chunk ch;
      ch.cap = bytes > ((size_t)256 << 20) ? bytes : ((size_t)256 << 20);
      if (dafs::hip_check(hipMalloc((void**)&ch.ptr, ch.cap))) return nullptr;
      dd_chunks.push_back(ch);
      best = dd_free_blocks.emplace(ch.ptr, ch.cap).first;
    }
    uint8_t* p = best->first;
    const size_t rest = best->second - bytes;
    dd_free_blocks.erase(best);
    if (rest) dd_free_blocks.emplace(p + bytes, rest);
    dd_in_use += bytes;
    if (dd_in_use > dd_peak) dd_peak = dd_in_use;
    return p;
  }
  void dd_reset() {
    dd_free_blocks.clear();
    for (dd_chunk& ch : dd_chunks) dd_free_blocks.emplace(ch.ptr, ch.cap);
    dd_in_use = 0;
    dd_open.clear();
    dd_wgs_in_flight[0] = dd_wgs_in_flight[1] = 0;
    dd_demotions = 0;
  }
  void dd_release() { for (dd_chunk& ch : dd_chunks) (void)hipFree(ch.ptr); dd_chunks.clear(); dd_free_blocks.clear(); dd_in_use = 0; dd_open.clear(); }
  uint32_t max_len() const { uint32_t m = 0; for (uint32_t l : len) m = l > m ? l : m; return m; }

  void free_all() {
    codes.release(); d_len.release(); d_seq_rp_off.release(); tasks.release(); scratch.release(); task_sim.release();
    counters.release(); d_sim.release(); d_pair_x.release(); d_pair_y.release(); mp_ent2.release(); mp_ident2.release(); pct_tasks.release(); work.release(); work2.release(); d_nodes.release(); d_paused.release(); d_nodes2.release(); d_paused2.release(); d_tref.release(); for (int k = 0; k < 2; ++k) { d_pack_off[k].release(); d_pack[k].release(); } dd_release();
    for (int k = 0; k < 2; ++k) { if (h_paused[k]) (void)hipHostFree(h_paused[k]); h_paused[k] = nullptr; h_paused_cap[k] = 0; }
    d_cf_params.release(); cf_seqs.release(); cf_codes.release(); cf_iws.release(); cf_cons.release(); cf_fws.release(); cf_post.release(); cf_logz.release();
    for (int k = 0; k < 2; ++k) { mp[k].release(); bp[k].release(); }
    bp_rows.release();
  }
};

// capi_align.cpp: similarity scores of all pairs from the rows of a store in row-major pair order (host + device sim)
int dafs_recompute_sim(dafs_hip_ctx* c, dafs::mp_store& st);
// capi_fold.cpp: CONTRAfold posteriors of the context's sequences seq[0..n) under constraint strings ("?.()", one per row),
// compacted with threshold th into `out`, whose "sequence" index is the row (Fold::Model::calculate(seq, str, bp), fold.cpp:191-207)
int dafs_fold_rows_constrained(dafs_hip_ctx* c, uint32_t n, const uint32_t* seq, const std::vector<std::string>& cons, float th, dafs::bp_store& out);
