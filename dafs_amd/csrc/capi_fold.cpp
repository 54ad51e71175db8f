// dafs_amd/csrc/capi_fold.cpp -- L1: base-pairing posteriors with the CONTRAfold model
// (Fold::Model::calculate, reference src/fold.cpp:60-68, 174-207).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <string>
#include <vector>

#include "../../include/dafs_hip.h"
#include "contrafold.h"
#include "contrafold_params.h"
#include "ctx.h"
#include "hip_util.h"

namespace dafs {

// LoadValues + InitializeCache (reference src/contrafold/InferenceEngine.ipp:1385-1397, 1106-1335):
// prefix sums of the *_at_least families, in float, in the reference's order.
void contrafold_default_params(cf_params* p) {
  memcpy(p->base_pair, cf_base_pair, sizeof p->base_pair);
  memcpy(p->terminal_mismatch, cf_terminal_mismatch, sizeof p->terminal_mismatch);
  memcpy(p->helix_stacking, cf_helix_stacking, sizeof p->helix_stacking);
  memcpy(p->helix_closing, cf_helix_closing, sizeof p->helix_closing);
  memcpy(p->dangle_left, cf_dangle_left, sizeof p->dangle_left);
  memcpy(p->dangle_right, cf_dangle_right, sizeof p->dangle_right);
  memcpy(p->bulge_0x1, cf_bulge_0x1_nucleotides, sizeof p->bulge_0x1);
  memcpy(p->bulge_1x0, cf_bulge_1x0_nucleotides, sizeof p->bulge_1x0);
  memcpy(p->internal_1x1, cf_internal_1x1_nucleotides, sizeof p->internal_1x1);
  p->multi_base = cf_multi_base; p->multi_unpaired = cf_multi_unpaired; p->multi_paired = cf_multi_paired;
  p->external_unpaired = cf_external_unpaired; p->external_paired = cf_external_paired;
  p->cache_hairpin[0] = cf_hairpin_length_at_least[0];
  for (int i = 1; i <= 30; i++) p->cache_hairpin[i] = p->cache_hairpin[i - 1] + cf_hairpin_length_at_least[i];
  float bulge[31], internal[31], sym[16], asym[29];
  bulge[0] = cf_bulge_length_at_least[0];
  for (int i = 1; i <= 30; i++) bulge[i] = bulge[i - 1] + cf_bulge_length_at_least[i];
  internal[0] = cf_internal_length_at_least[0];
  for (int i = 1; i <= 30; i++) internal[i] = internal[i - 1] + cf_internal_length_at_least[i];
  sym[0] = cf_internal_symmetric_length_at_least[0];
  for (int i = 1; i <= 15; i++) sym[i] = sym[i - 1] + cf_internal_symmetric_length_at_least[i];
  asym[0] = cf_internal_asymmetry_at_least[0];
  for (int i = 1; i <= 28; i++) asym[i] = asym[i - 1] + cf_internal_asymmetry_at_least[i];
  for (int l1 = 0; l1 <= 30; l1++)
    for (int l2 = 0; l2 <= 30; l2++) {
      float v = 0.0f;
      if (l1 + l2 <= 30 && !(l1 == 0 && l2 == 0)) {
        if (l1 == 0 || l2 == 0) {
          v += bulge[std::min(30, l1 + l2)];
        } else {
          if (l1 <= 4 && l2 <= 4) v += cf_internal_explicit[l1][l2];
          v += internal[std::min(30, l1 + l2)];
          if (l1 == l2) v += sym[std::min(15, l1)];
          v += asym[std::min(28, l1 > l2 ? l1 - l2 : l2 - l1)];
        }
      }
      p->cache_single[l1 * 31 + l2] = v;
    }
}

}  // namespace dafs

using namespace dafs;

namespace {

// SStruct::ConvertParensToMapping (reference src/contrafold/SStruct.cpp:389-417; '-' reads as '.')
int parse_constraint(const char* cons, uint32_t L, std::vector<int>& map) {
  map.assign(L + 1, -1);
  std::vector<int> stack;
  for (uint32_t i = 1; i <= L; ++i) {
    const char ch = cons[i - 1];
    if (ch == '?') continue;
    if (ch == '.' || ch == '-') map[i] = 0;
    else if (ch == '(') stack.push_back((int)i);
    else if (ch == ')') {
      if (stack.empty()) return DAFS_HIP_EINVAL;
      map[i] = stack.back();
      map[stack.back()] = (int)i;
      stack.pop_back();
    } else return DAFS_HIP_EINVAL;
  }
  return stack.empty() ? DAFS_HIP_OK : DAFS_HIP_EINVAL;
}

struct fold_job {
  std::vector<cf_seq> seqs;
  uint64_t iws = 0, fws = 0, post = 0;
  void add(uint32_t len, uint32_t code_off, bool has_cons, uint32_t cons_off) {
    cf_seq s;
    memset(&s, 0, sizeof s);
    s.len = len; s.code_off = code_off; s.has_constraint = has_cons ? 1u : 0u; s.cons_off = cons_off;
    const uint64_t S = (uint64_t)(len + 1) * (len + 2) / 2;
    s.iws_off = iws; iws += 12ull * (len + 2);
    s.fws_off = fws; fws += 7 * S + 2ull * (len + 1);
    s.post_off = post; post += S;
    seqs.push_back(s);
  }
};

int ensure_params(dafs_hip_ctx* c, hipStream_t st) {
  if (c->cf_params_ready) return DAFS_HIP_OK;
  cf_params p;
  contrafold_default_params(&p);
  int rc = c->d_cf_params.upload((const uint8_t*)&p, sizeof p, st);
  if (rc) return rc;
  c->cf_params_ready = true;
  return DAFS_HIP_OK;
}

int run_job(dafs_hip_ctx* c, const fold_job& job, const uint8_t* d_codes, const int* d_cons, cf_batch* out, hipStream_t st) {
  int rc;
  if ((rc = ensure_params(c, st))) return rc;
  if ((rc = c->cf_seqs.upload((const uint8_t*)job.seqs.data(), job.seqs.size() * sizeof(cf_seq), st))) return rc;
  if ((rc = c->cf_iws.reserve(job.iws))) return rc;
  if ((rc = c->cf_fws.reserve(job.fws))) return rc;
  if ((rc = c->cf_post.reserve(job.post))) return rc;
  if ((rc = c->cf_logz.reserve(job.seqs.size()))) return rc;
  cf_batch B;
  B.params = (const cf_params*)c->d_cf_params.ptr;
  B.seqs = (const cf_seq*)c->cf_seqs.ptr;
  B.codes = d_codes;
  B.cons = d_cons;
  B.iws = c->cf_iws.ptr;
  B.fws = c->cf_fws.ptr;
  B.post = c->cf_post.ptr;
  B.logz = c->cf_logz.ptr;
  B.stamps = nullptr;
  if (getenv("DAFS_HIP_CF_STAMPS")) {
    if ((rc = c->cf_stamps.reserve(8))) return rc;
    B.stamps = c->cf_stamps.ptr;
  }
  *out = B;
  uint32_t max_len = 0;
  for (const cf_seq& q : job.seqs) max_len = std::max(max_len, q.len);
  rc = contrafold_launch(B, (uint32_t)job.seqs.size(), max_len, st);
  if (!rc && B.stamps) {
    unsigned long long h[8] = {0};
    if (!hip_check(hipMemcpy(h, B.stamps, sizeof h, hipMemcpyDeviceToHost)))
      fprintf(stderr, "k_contrafold block 0 (us): inside %.0f (terms %.0f) | F5i %.0f | F5o %.0f | outside %.0f (terms %.0f)\n", (h[1] - h[0]) / 100.0,
              h[5] / 100.0, (h[2] - h[1]) / 100.0, (h[3] - h[2]) / 100.0, (h[4] - h[3]) / 100.0, h[6] / 100.0);
  }
  return rc;
}

}  // namespace

// Batch hook: Fold::Model::calculate(const vector<Fasta>&, vector<BP>&) for -s CONTRAfold.
// Result: the context's un-relaxed base-pairing store (rows with p > th; reference CUTOFF 0.01).
// _begin enqueues the inside/outside/posterior kernels on the context's folding stream and returns; the pair
// posteriors and the matching-probability transform do not depend on them and may run meanwhile.  _end waits
// for the kernels and compacts the posteriors into the store.
extern "C" int dafs_hip_fold_posteriors_begin(dafs_hip_ctx* c, int model, float th) {
  if (!c || c->len.empty() || model != DAFS_FOLD_CONTRAFOLD) return DAFS_HIP_EINVAL;
  if (hip_check(hipSetDevice(c->device))) return DAFS_HIP_ENODEV;
  if (c->fold_pending) return DAFS_HIP_EINVAL;
  const uint32_t n = (uint32_t)c->len.size();
  fold_job job;
  for (uint32_t x = 0; x < n; ++x) job.add(c->len[x], c->off[x], false, 0);
  cf_batch B;
  int rc = run_job(c, job, c->codes.ptr, nullptr, &B, c->fold_stream);
  if (rc) return rc;
  c->fold_batch.assign((const uint8_t*)&B, (const uint8_t*)&B + sizeof B);
  c->fold_pending = true;
  c->fold_th = th;
  c->bp[0].valid = false;
  c->bp[1].valid = false;
  c->cur_bp = 0;
  return DAFS_HIP_OK;
}

extern "C" int dafs_hip_fold_posteriors_end(dafs_hip_ctx* c) {
  if (!c || !c->fold_pending || c->fold_batch.size() != sizeof(cf_batch)) return DAFS_HIP_EINVAL;
  if (hip_check(hipSetDevice(c->device))) return DAFS_HIP_ENODEV;
  c->fold_pending = false;
  if (hip_check(hipStreamSynchronize(c->fold_stream))) return DAFS_HIP_ELAUNCH;
  cf_batch B;
  memcpy(&B, c->fold_batch.data(), sizeof B);
  const float th = c->fold_th;
  const uint32_t n = (uint32_t)c->len.size();
  int rc;
  bp_store& st = c->bp[0];
  if ((rc = st.rowptr.reserve(c->seq_rp_off[n]))) return rc;
  if ((rc = st.nnz.reserve(n))) return rc;
  if ((rc = st.bp_off.reserve(n + 1))) return rc;
  if ((rc = st.rp_off.upload(c->seq_rp_off.data(), n + 1, c->stream))) return rc;
  if ((rc = c->counters.reserve(4))) return rc;
  uint64_t cap = 8ull * c->off[n] + 1024;
  for (int attempt = 0;; ++attempt) {
    if ((rc = st.col.reserve(cap))) return rc;
    if ((rc = st.val.reserve(cap))) return rc;
    if (hip_check(hipMemsetAsync(c->counters.ptr, 0, 4 * sizeof(unsigned long long), c->stream))) return DAFS_HIP_ELAUNCH;
    if ((rc = bp_compact_launch(B, n, th, st.rp_off.ptr, st.rowptr.ptr, st.col.ptr, st.val.ptr, st.bp_off.ptr, st.nnz.ptr,
                                c->counters.ptr, cap, (int*)(c->counters.ptr + 2), c->stream)))
      return rc;
    unsigned long long h[4];
    if (hip_check(hipMemcpyAsync(h, c->counters.ptr, sizeof h, hipMemcpyDeviceToHost, c->stream))) return DAFS_HIP_ELAUNCH;
    if (hip_check(hipStreamSynchronize(c->stream))) return DAFS_HIP_ELAUNCH;
    const int status = (int)(h[2] & 0xffffffffu);
    if (status == 0) { st.total_nnz = h[0]; st.valid = true; return DAFS_HIP_OK; }
    if (status != DAFS_HIP_EOVERFLOW || attempt >= 4) return status;
    cap = std::max<uint64_t>(h[0], cap * 2);
  }
}

extern "C" int dafs_hip_fold_posteriors(dafs_hip_ctx* c, int model, float th) {
  const int rc = dafs_hip_fold_posteriors_begin(c, model, th);
  return rc ? rc : dafs_hip_fold_posteriors_end(c);
}

// Constrained posteriors of the rows of one alignment (DAFS::update_basepairing_probability, dafs.cpp:657-663: one
// s_model_->calculate(seq, con, bp) per sequence), as one batch; the result is a store whose index is the row.
int dafs_fold_rows_constrained(dafs_hip_ctx* c, uint32_t n, const uint32_t* seq, const std::vector<std::string>& cons, float th, dafs::bp_store& st) {
  if (!c || !n || !seq || cons.size() != n || c->fold_pending) return DAFS_HIP_EINVAL;
  st.valid = false;
  fold_job job;
  std::vector<int> maps, one;
  std::vector<uint64_t> rp_off(n + 1, 0);
  uint64_t residues = 0;
  for (uint32_t r = 0; r < n; ++r) {
    if (seq[r] >= c->len.size()) return DAFS_HIP_EINVAL;
    const uint32_t L = c->len[seq[r]];
    if (cons[r].size() < L) return DAFS_HIP_EINVAL;
    int rc = parse_constraint(cons[r].c_str(), L, one);
    if (rc) return rc;
    job.add(L, c->off[seq[r]], true, (uint32_t)maps.size());
    maps.insert(maps.end(), one.begin(), one.end());
    rp_off[r + 1] = rp_off[r] + L + 1;
    residues += L;
  }
  int rc;
  if ((rc = c->cf_cons.upload(maps.data(), maps.size(), c->stream))) return rc;
  cf_batch B;
  if ((rc = run_job(c, job, c->codes.ptr, c->cf_cons.ptr, &B, c->stream))) return rc;
  if ((rc = st.rowptr.reserve(rp_off[n]))) return rc;
  if ((rc = st.nnz.reserve(n))) return rc;
  if ((rc = st.bp_off.reserve(n + 1))) return rc;
  if ((rc = st.rp_off.upload(rp_off.data(), n + 1, c->stream))) return rc;
  if ((rc = c->counters.reserve(4))) return rc;
  uint64_t cap = 8ull * residues + 1024;
  for (int attempt = 0;; ++attempt) {
    if ((rc = st.col.reserve(cap))) return rc;
    if ((rc = st.val.reserve(cap))) return rc;
    if (hip_check(hipMemsetAsync(c->counters.ptr, 0, 4 * sizeof(unsigned long long), c->stream))) return DAFS_HIP_ELAUNCH;
    if ((rc = bp_compact_launch(B, n, th, st.rp_off.ptr, st.rowptr.ptr, st.col.ptr, st.val.ptr, st.bp_off.ptr, st.nnz.ptr,
                                c->counters.ptr, cap, (int*)(c->counters.ptr + 2), c->stream)))
      return rc;
    unsigned long long h[4];
    if (hip_check(hipMemcpyAsync(h, c->counters.ptr, sizeof h, hipMemcpyDeviceToHost, c->stream))) return DAFS_HIP_ELAUNCH;
    if (hip_check(hipStreamSynchronize(c->stream))) return DAFS_HIP_ELAUNCH;
    const int status = (int)(h[2] & 0xffffffffu);
    if (status == 0) { st.total_nnz = h[0]; st.valid = true; return DAFS_HIP_OK; }
    if (status != DAFS_HIP_EOVERFLOW || attempt >= 4) return status;
    cap = std::max<uint64_t>(h[0], cap * 2);
  }
}

// Single-sequence plugin call: CONTRAfold<float>::ComputePosterior (reference
// src/contrafold/wrapper.cpp:181-200), optionally under a constraint string of len chars from
// "?.()" (Fold::Model::calculate(seq, str, bp), src/fold.cpp:191-207).  post receives the
// (len+1)(len+2)/2 triangular posteriors in the reference's layout.
extern "C" int dafs_hip_fold_posterior_dense(dafs_hip_ctx* c, const char* seq, uint32_t len, const char* constraint, float* post,
                                             float* logz) {
  if (!c || !seq || !len || !post) return DAFS_HIP_EINVAL;
  if (hip_check(hipSetDevice(c->device))) return DAFS_HIP_ENODEV;
  if (constraint && !constraint[0]) constraint = nullptr;  // empty = unconstrained (wrapper.cpp:188)
  std::vector<uint8_t> codes(len);
  for (uint32_t i = 0; i < len; ++i) codes[i] = dafs_hip_residue_code(seq[i]);
  std::vector<int> map;
  if (constraint) {
    if (strlen(constraint) < len) return DAFS_HIP_EINVAL;
    int rc = parse_constraint(constraint, len, map);
    if (rc) return rc;
  }
  int rc;
  if ((rc = c->cf_codes.upload(codes.data(), len, c->stream))) return rc;
  if (constraint && (rc = c->cf_cons.upload(map.data(), map.size(), c->stream))) return rc;
  fold_job job;
  job.add(len, 0, constraint != nullptr, 0);
  cf_batch B;
  if (c->fold_pending) return DAFS_HIP_EINVAL;  // the batch in flight owns the folding workspaces
  if ((rc = run_job(c, job, c->cf_codes.ptr, constraint ? c->cf_cons.ptr : nullptr, &B, c->stream))) return rc;
  const uint64_t S = (uint64_t)(len + 1) * (len + 2) / 2;
  if (hip_check(hipMemcpyAsync(post, B.post, S * 4, hipMemcpyDeviceToHost, c->stream))) return DAFS_HIP_ELAUNCH;
  float z = 0;
  if (hip_check(hipMemcpyAsync(&z, B.logz, 4, hipMemcpyDeviceToHost, c->stream))) return DAFS_HIP_ELAUNCH;
  if (hip_check(hipStreamSynchronize(c->stream))) return DAFS_HIP_ELAUNCH;
  if (logz) *logz = z;
  return DAFS_HIP_OK;
}
