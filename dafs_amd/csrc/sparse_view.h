// dafs_amd/csrc/sparse_view.h -- device-side access to the sparse posterior stores.
//
// MP store: one entry per unordered sequence pair {a<b} ("task"), holding the row lists of
// mp[a][b] and, right behind them, of mp[b][a] (the layout k_pairhmm3 writes, include/dafs_hip.h).
// mp_rows(a,b,i) returns row i of mp[a][b] for any a != b; mp[a][a] is the identity
// (reference src/align.cpp:42-44) and is handled by the callers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dafs {

struct mp_store_dev {
  const uint32_t* rowptr_pool;
  const uint32_t* col;
  const float* val;
  const uint2* ent2;             // optional: the same entries interleaved, {col, bits of val} (the consistency kernels' gathers
                                 // fetch one line per row instead of two; built per transform by pct_interleave_launch)
  const uint2* ident2;           // optional, with ent2: ident2[k] = {k, bits of 1.0f}, the row k of an identity matrix mp[x][x]
  const uint32_t* ident_rp;      // optional, with ident2: ident_rp[k] = k, the row pointers of that identity matrix
  const uint64_t* pair_off;      // per task: first entry of mp[a][b]; mp[b][a] follows after nnz entries
  const uint32_t* pair_nnz;      // per task
  const uint64_t* rp_off;        // per task: first row pointer (len[a]+1 of them, then len[b]+1)
  const uint32_t* task_of_pair;  // row-major pair id (a<b) -> task
  const uint32_t* len;           // sequence lengths
  uint32_t nseq;
};

struct bp_store_dev {
  const uint32_t* rowptr;  // concatenated, per sequence len+1 entries, relative to bp_off[x]
  const uint32_t* col;
  const float* val;
  const uint64_t* rp_off;  // per sequence: first row pointer
  const uint64_t* bp_off;  // per sequence: first entry
};

struct row_ref {
  const uint32_t* col;
  const float* val;
  uint32_t n;
};

__device__ __forceinline__ uint32_t pair_id(uint32_t a, uint32_t b, uint32_t n) {  // a < b
  return a * n - a * (a + 1) / 2 + (b - a - 1);
}

// row i of mp[a][b], a != b
__device__ __forceinline__ row_ref mp_row(const mp_store_dev& s, uint32_t a, uint32_t b, uint32_t i) {
  row_ref r;
  if (a < b) {
    const uint32_t t = s.task_of_pair[pair_id(a, b, s.nseq)];
    const uint32_t* rp = s.rowptr_pool + s.rp_off[t];
    const uint64_t base = s.pair_off[t];
    const uint32_t beg = rp[i];
    r.n = rp[i + 1] - beg;
    r.col = s.col + base + beg;
    r.val = s.val + base + beg;
  } else {
    const uint32_t t = s.task_of_pair[pair_id(b, a, s.nseq)];
    const uint32_t* rp = s.rowptr_pool + s.rp_off[t] + s.len[b] + 1;
    const uint64_t base = s.pair_off[t] + s.pair_nnz[t];
    const uint32_t beg = rp[i];
    r.n = rp[i + 1] - beg;
    r.col = s.col + base + beg;
    r.val = s.val + base + beg;
  }
  return r;
}

__device__ __forceinline__ row_ref bp_row(const bp_store_dev& s, uint32_t x, uint32_t i) {
  const uint32_t* rp = s.rowptr + s.rp_off[x];
  row_ref r;
  const uint32_t beg = rp[i];
  r.n = rp[i + 1] - beg;
  r.col = s.col + s.bp_off[x] + beg;
  r.val = s.val + s.bp_off[x] + beg;
  return r;
}

}  // namespace dafs
