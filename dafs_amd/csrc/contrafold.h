// dafs_amd/csrc/contrafold.h -- launch descriptors of the CONTRAfold kernels (contrafold.hip)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dafs {

// Physical score tables (symbol index 0..4, 4 = non-ACGU scores 0) plus the two caches
// InferenceEngine::InitializeCache derives (reference src/contrafold/InferenceEngine.ipp:1106-1335).
struct cf_params {
  float base_pair[25];
  float terminal_mismatch[625];
  float helix_stacking[625];
  float helix_closing[25];
  float dangle_left[125];
  float dangle_right[125];
  float bulge_0x1[5];
  float bulge_1x0[5];
  float internal_1x1[25];
  float cache_hairpin[31];
  float cache_single[31 * 31];
  float multi_base, multi_unpaired, multi_paired, external_unpaired, external_paired;
};

struct cf_seq {
  uint32_t len;
  uint32_t code_off;       // into cf_batch.codes
  uint32_t has_constraint;
  uint32_t cons_off;       // into cf_batch.cons (len+1 ints, index = position)
  uint64_t iws_off;        // into cf_batch.iws: 12*(len+2) ints (integer side tables, written by k_contrafold)
  uint64_t fws_off;        // into cf_batch.fws: 7*S + 2*(len+1) floats, S = (len+1)(len+2)/2
  uint64_t post_off;       // into cf_batch.post: S floats
};

struct cf_batch {
  const cf_params* params;  // [device]
  const cf_seq* seqs;       // [device]
  const uint8_t* codes;     // residue class codes (ProbCons classes; A C G U = 0..3)
  const int* cons;          // constraint mappings (-1 unknown, 0 unpaired, else partner position)
  int* iws;
  float* fws;
  float* post;              // triangular posteriors, reference layout
  float* logz;              // optional [nseq]
  unsigned long long* stamps;  // optional [8]: 100 MHz timestamps of block 0 at phase boundaries (tuning aid)
};

void contrafold_default_params(cf_params* p);  // host: tables + caches
int contrafold_launch(const cf_batch& B, uint32_t nseq, uint32_t max_len, hipStream_t st);
int bp_compact_launch(const cf_batch& B, uint32_t nseq, float th, const uint64_t* rp_off, uint32_t* out_rowptr, uint32_t* out_col,
                      float* out_val, uint64_t* out_off, uint32_t* out_nnz, unsigned long long* pool_top, uint64_t pool_cap, int* status,
                      hipStream_t st);

}  // namespace dafs
