// dafs_amd/csrc/pct.h -- launch arguments of the consistency-transform kernels (pct.hip)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <vector>
#include "sparse_view.h"

namespace dafs {

struct pct_match_args {
  mp_store_dev in;        // un-relaxed matching probabilities
  const float* sim;       // N*N similarity matrix (diagonal 1)
  const uint32_t* pair_x; // output pair p -> (x, y), row-major x<y
  const uint32_t* pair_y;
  uint32_t npairs;
  float w_pct;            // reference -p (w_pct_a_)
  // output store; task == row-major pair id
  const uint64_t* rp_off;
  uint32_t* rowptr_pool;
  uint32_t* col;
  float* val;
  unsigned long long* pool_top;
  uint64_t pool_cap;
  uint64_t* pair_off;
  uint32_t* pair_nnz;
  int* status;
  // dense row tiles of the pairs of one launch: tile + tile_off[k] is the L1 x L2 tile of the launch's k-th pair
  float* tile;
  const uint64_t* tile_off;
  float* sum_w;           // per pair of the launch: sum of the weights w_z (written by the row kernel)
  uint32_t max_len;       // filled by the launcher
  // optional (k_pct_rows): the launch's workgroups as a 1-D list of {pair index within the launch, block of 16 rows}; null =
  // a 2-D grid (pair, row block).  pct_task_order builds it.
  const uint2* wg_task;
  uint32_t wg_tasks;
  // four-way transform only (pct_fourway_launch): the un-relaxed base-pairing store and the weight -f
  bp_store_dev bp;
  float w_f;
};

struct pct_bp_args {
  mp_store_dev mp;    // un-relaxed matching probabilities
  bp_store_dev bp;    // un-relaxed base-pairing probabilities
  const float* sim;
  float w_pct;        // reference -q (w_pct_s_)
  // output: same row-pointer layout as the input store (bp.rp_off)
  uint32_t* out_rowptr;
  uint32_t* out_col;
  float* out_val;
  unsigned long long* pool_top;
  uint64_t pool_cap;
  uint64_t* out_off;
  uint32_t* out_nnz;
  int* status;
  // dense L x L tiles, one per sequence (tile + tile_off[x]), and the sums of the weights w_y
  float* tile;
  const uint64_t* tile_off;
  float* sum_w;
  uint32_t max_len;
};

// Workgroup order of k_pct_rows for the pairs [p0, p0 + count) of a launch (host): the tasks (pair, block of 16 rows) sorted by
// (y, row block, x) -- every workgroup of such a run gathers the same b-rows mp[z][y][k ~ row block], only the a-side differs --
// and dealt to the XCDs in contiguous ranges (workgroup b runs on XCD b mod 8), so that what an XCD's L2 (4 MB) serves at a
// time is one or two (y, row block) groups instead of a slice of every matrix of the store.  Padding entries hold x = ~0.
void pct_task_order(const uint32_t* pair_x, const uint32_t* pair_y, const uint32_t* len, uint32_t nseq, uint64_t p0, uint32_t count,
                    std::vector<uint2>& out);

// ent2[e] = {col[e], bits of val[e]} for e < n: the interleaved copy the row kernels gather from (mp_store_dev::ent2)
int pct_interleave_launch(const uint32_t* col, const float* val, uint2* ent2, uint64_t n, hipStream_t st);
int pct_ident_launch(uint2* ident2, uint32_t* ident_rp, uint32_t n, hipStream_t st);  // ident2[k] = {k, bits of 1.0f}, ident_rp[k] = k for k < n
int pct_match_launch(pct_match_args a, uint32_t max_len, uint32_t pair0, uint32_t count, hipStream_t st);
int pct_bp_launch(pct_bp_args a, uint32_t max_len, hipStream_t st);
// DAFS::relax_fourway_consistency (dafs.cpp:377-444) for the pairs [pair0, pair0 + count): same outputs as pct_match_launch
int pct_fourway_launch(pct_match_args a, uint32_t max_len, uint32_t pair0, uint32_t count, hipStream_t st);
// similarity scores of all pairs from a stored matching-probability set; row_dp / row_tr: (max_len + 1) * npairs words each
int mp_sim_launch(mp_store_dev in, const uint32_t* pair_x, const uint32_t* pair_y, uint32_t npairs, float* task_sim, float* row_dp, int* row_tr,
                  hipStream_t st);

}  // namespace dafs
