// dafs_amd/csrc/capi_shard.cpp -- dafs_hip_phase1_sharded: phase 1 of DAFS::run (reference src/dafs.cpp:1787-1827) on one
// rank of a multi-GPU run, with the caller's all-gather between the pieces.
//
// What is independent in the reference and therefore sharded here, one process per GPU:
//   * the N base-pairing matrices (Fold::Model::calculate loops over the sequences, src/fold.cpp:66-67): rank r folds the
//     sequences x with x mod world == r in a context of its own;
//   * the N(N-1)/2 pair posteriors (Align::Model::calculate, src/align.cpp:46-50): rank r computes a contiguous range of the
//     row-major pair enumeration, so the ranks' ranges in rank order ARE the whole in pair order;
//   * relax_matching_probability's output pairs (src/dafs.cpp:265-315): the same ranges.
// relax_basepairing_probability (:326-375) costs milliseconds and is replicated.  Between the pieces the stores travel as
// device buffers (capi_dev.cpp): sizes first, then ONE all-gather of a packed, max-padded slab per exchange.  The library
// knows nothing about the transport: `allgather` is RCCL in dafs_amd/csrc/host/cli_main.cpp (dafs --devices).
#include <hip/hip_runtime.h>
#include <string.h>
#include <algorithm>
#include <vector>

#include "../../include/dafs_hip.h"
#include "ctx.h"
#include "hip_util.h"

using namespace dafs;

namespace {

struct dbuf {  // device memory of one exchange, freed on scope exit
  void* p = nullptr;
  ~dbuf() { if (p) (void)hipFree(p); }
  int alloc(size_t bytes) { return hip_check(hipMalloc(&p, bytes ? bytes : 4)) ? DAFS_HIP_ENOMEM : DAFS_HIP_OK; }
  template <class T> T* as() const { return (T*)p; }
};

struct part { const void* ptr; uint64_t words; };  // one 4-byte array of this rank's contribution

// Every rank contributes parts.size() arrays (the same number everywhere, lengths differ); out[k] receives the concatenation
// over the ranks of part k (device memory owned by `store`), n_out[k] its length in words.
int gather_parts(dafs_hip_ctx* c, uint32_t world, const std::vector<part>& parts, dafs_allgather_fn ag, void* user, dbuf& store,
                 std::vector<uint32_t*>& out, std::vector<uint64_t>& n_out) {
  const size_t np = parts.size();
  // the strides have to be agreed: one tiny all-gather of the lengths
  std::vector<uint64_t> mine(np), all((size_t)world * np);
  for (size_t k = 0; k < np; ++k) mine[k] = parts[k].words;
  dbuf sz_send, sz_recv;
  int rc;
  if ((rc = sz_send.alloc(np * 8)) || (rc = sz_recv.alloc((size_t)world * np * 8))) return rc;
  if (hip_check(hipMemcpyAsync(sz_send.p, mine.data(), np * 8, hipMemcpyHostToDevice, c->stream))) return DAFS_HIP_ELAUNCH;
  if (hip_check(hipStreamSynchronize(c->stream))) return DAFS_HIP_ELAUNCH;
  if (ag(user, sz_send.p, sz_recv.p, np * 8, (void*)c->stream)) return DAFS_HIP_ECOMM;
  if (hip_check(hipStreamSynchronize(c->stream))) return DAFS_HIP_ELAUNCH;
  if (hip_check(hipMemcpy(all.data(), sz_recv.p, (size_t)world * np * 8, hipMemcpyDeviceToHost))) return DAFS_HIP_ELAUNCH;
  std::vector<uint64_t> stride(np, 0), off(np + 1, 0), total(np, 0);
  for (uint32_t r = 0; r < world; ++r)
    for (size_t k = 0; k < np; ++k) { stride[k] = std::max(stride[k], all[r * np + k]); total[k] += all[r * np + k]; }
  for (size_t k = 0; k < np; ++k) off[k + 1] = off[k] + stride[k];
  const uint64_t slab = off[np];  // words per rank
  out.assign(np, nullptr);
  n_out = total;
  if (slab == 0) return DAFS_HIP_OK;
  dbuf send, recv;
  if ((rc = send.alloc(slab * 4)) || (rc = recv.alloc((size_t)world * slab * 4))) return rc;
  for (size_t k = 0; k < np; ++k)
    if (parts[k].words && hip_check(hipMemcpyAsync(send.as<uint32_t>() + off[k], parts[k].ptr, parts[k].words * 4, hipMemcpyDeviceToDevice, c->stream)))
      return DAFS_HIP_ELAUNCH;
  if (hip_check(hipStreamSynchronize(c->stream))) return DAFS_HIP_ELAUNCH;
  if (ag(user, send.p, recv.p, slab * 4, (void*)c->stream)) return DAFS_HIP_ECOMM;
  // the padding comes out here: part k of rank r sits at recv[r * slab + off[k]]
  uint64_t sum = 0;
  for (size_t k = 0; k < np; ++k) sum += total[k];
  if ((rc = store.alloc(sum * 4))) return rc;
  uint64_t at = 0;
  for (size_t k = 0; k < np; ++k) {
    out[k] = store.as<uint32_t>() + at;
    for (uint32_t r = 0; r < world; ++r) {
      const uint64_t n = all[r * np + k];
      if (n && hip_check(hipMemcpyAsync(store.as<uint32_t>() + at, recv.as<uint32_t>() + (size_t)r * slab + off[k], n * 4, hipMemcpyDeviceToDevice, c->stream)))
        return DAFS_HIP_ELAUNCH;
      at += n;
    }
  }
  if (hip_check(hipStreamSynchronize(c->stream))) return DAFS_HIP_ELAUNCH;
  return DAFS_HIP_OK;
}

}  // namespace

extern "C" void dafs_hip_pair_range(uint64_t npairs, uint32_t world, uint32_t rank, uint64_t* begin, uint64_t* end) {
  // contiguous ranges of near-equal pair counts, the same split dafs_amd/dist.py::pair_ranges makes
  const uint64_t w = world ? world : 1;
  if (begin) *begin = npairs * rank / w;
  if (end) *end = npairs * ((uint64_t)rank + 1) / w;
}

extern "C" int dafs_hip_phase1_sharded(dafs_hip_ctx* c, uint32_t rank, uint32_t world, int align_model, float th_a, float w_pct_a, float w_pct_s,
                                       int fold_model, float fold_th, dafs_allgather_fn allgather, void* user) {
  if (!c || !allgather || world == 0 || rank >= world || c->len.size() < 2) return DAFS_HIP_EINVAL;
  if (hip_check(hipSetDevice(c->device))) return DAFS_HIP_ENODEV;
  const uint32_t n = (uint32_t)c->len.size();
  const uint64_t npairs = (uint64_t)n * (n - 1) / 2;
  int rc;
  std::vector<uint32_t*> g;
  std::vector<uint64_t> gn;

  {  // ---- base-pairing posteriors of the sequences x = rank (mod world), folded in a context of their own ----
    std::vector<const char*> seqs;
    std::vector<uint32_t> lens;
    for (uint32_t x = rank; x < n; x += world) { seqs.push_back(c->seq.data() + c->off[x]); lens.push_back(c->len[x]); }
    dbuf rp, col, val, store;
    uint64_t n_rp = 0, n_ent = 0;
    dafs_hip_ctx* fc = nullptr;
    if (!seqs.empty()) {
      if ((rc = dafs_hip_create(c->device, &fc))) return rc;
      struct closer { dafs_hip_ctx* f; ~closer() { dafs_hip_destroy(f); } } guard{fc};
      uint64_t nnz = 0, nrp = 0;
      if ((rc = dafs_hip_set_sequences(fc, (uint32_t)seqs.size(), seqs.data(), lens.data())) || (rc = dafs_hip_fold_posteriors(fc, fold_model, fold_th)) ||
          (rc = dafs_hip_bp_result_size(fc, 0, &nnz, &nrp)))
        return rc;
      if ((rc = rp.alloc(nrp * 4)) || (rc = col.alloc(nnz * 4)) || (rc = val.alloc(nnz * 4))) return rc;
      if ((rc = dafs_hip_bp_export_dev(fc, rp.as<uint32_t>(), col.as<uint32_t>(), val.as<float>(), nnz, &n_rp, &n_ent))) return rc;
    }
    if ((rc = gather_parts(c, world, {{rp.p, n_rp}, {col.p, n_ent}, {val.p, n_ent}}, allgather, user, store, g, gn))) return rc;
    std::vector<uint32_t> order;  // sequence of the k-th gathered block
    for (uint32_t r = 0; r < world; ++r)
      for (uint32_t x = r; x < n; x += world) order.push_back(x);
    if ((rc = dafs_hip_set_bp_dev(c, n, order.data(), g[0], g[1], (const float*)g[2], gn[1]))) return rc;
  }

  uint64_t b0, b1;
  dafs_hip_pair_range(npairs, world, rank, &b0, &b1);
  const uint64_t cnt = b1 - b0;
  {  // ---- pair posteriors + similarity scores of this rank's pair range ----
    dbuf nnz, rp, col, val, sim, store;
    uint64_t n_rp = 0, n_ent = 0;
    if (cnt) {
      uint64_t np_ = 0, nz = 0, nrp = 0;
      if ((rc = dafs_hip_align_posteriors(c, align_model, th_a, b0, b1)) || (rc = dafs_hip_mp_result_size(c, 0, &np_, &nz, &nrp))) return rc;
      if ((rc = nnz.alloc(cnt * 4)) || (rc = rp.alloc(nrp * 4)) || (rc = col.alloc(nz * 8)) || (rc = val.alloc(nz * 8)) || (rc = sim.alloc(cnt * 4))) return rc;
      if ((rc = dafs_hip_mp_export_dev(c, 0, 0, cnt, nnz.as<uint32_t>(), rp.as<uint32_t>(), col.as<uint32_t>(), val.as<float>(), sim.as<float>(), 2 * nz, &n_rp,
                                       &n_ent)))
        return rc;
    }
    if ((rc = gather_parts(c, world, {{nnz.p, cnt}, {rp.p, n_rp}, {col.p, n_ent}, {val.p, n_ent}, {sim.p, cnt}}, allgather, user, store, g, gn))) return rc;
    if ((rc = dafs_hip_mp_install_dev(c, 0, g[0], g[1], g[2], (const float*)g[3], (const float*)g[4], gn[2]))) return rc;
  }

  if (w_pct_a != 0.0f) {  // ---- relax_matching_probability for this rank's range of output pairs ----
    dbuf nnz, rp, col, val, store;
    uint64_t n_rp = 0, n_ent = 0;
    if (cnt) {
      uint64_t np_ = 0, nz = 0, nrp = 0;
      if ((rc = dafs_hip_consistency_match_range(c, w_pct_a, b0, b1)) || (rc = dafs_hip_mp_result_size(c, 1, &np_, &nz, &nrp))) return rc;
      if ((rc = nnz.alloc(cnt * 4)) || (rc = rp.alloc(nrp * 4)) || (rc = col.alloc(nz * 8)) || (rc = val.alloc(nz * 8))) return rc;
      if ((rc = dafs_hip_mp_export_dev(c, 1, b0, cnt, nnz.as<uint32_t>(), rp.as<uint32_t>(), col.as<uint32_t>(), val.as<float>(), nullptr, 2 * nz, &n_rp, &n_ent)))
        return rc;
    }
    if ((rc = gather_parts(c, world, {{nnz.p, cnt}, {rp.p, n_rp}, {col.p, n_ent}, {val.p, n_ent}}, allgather, user, store, g, gn))) return rc;
    if ((rc = dafs_hip_mp_install_dev(c, 1, g[0], g[1], g[2], (const float*)g[3], nullptr, gn[2]))) return rc;
  }
  if (w_pct_s != 0.0f && (rc = dafs_hip_consistency_bp(c, w_pct_s))) return rc;
  return DAFS_HIP_OK;
}
