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
#include "stage.h"

namespace dafs {

stage_recorder*& stage_current() {
  static stage_recorder* cur = nullptr;
  return cur;
}

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
    case DAFS_HIP_ECOMM: return "dafs_hip: the collective between the ranks failed";
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

// CONTRAlign RNA defaults (reference src/contralign/Defaults.ipp:389-419) expanded to the physical
// tables InferenceEngine::RegisterParameters ties them to (src/contralign/InferenceEngine.ipp:139-226).
extern "C" void dafs_hip_pairhmm5_default_model(dafs_pairhmm5_model* m) {
  enum { M = 0, IX = 1, IY = 2, I2X = 3, I2Y = 4 };
  memset(m, 0, sizeof *m);
  // match_XY, name = lexicographic min of XY / YX: AA AC AG AU CC CG CU GG GU UU
  static const float mt[10] = {(float)(0.5256508867), (float)(-0.4090640200), (float)(-0.2502759109), (float)(-0.3252306723),
                               (float)(0.6665219366), (float)(-0.3289391181), (float)(-0.1326088918), (float)(0.6684676551),
                               (float)(-0.3565888168), (float)(0.4590520450)};
  int t = 0;
  for (int i = 0; i < 4; ++i)
    for (int j = i; j < 4; ++j) m->match[i][j] = m->match[j][i] = mt[t++];
  static const float ins[4] = {(float)(-0.0025219272), (float)(-0.0831389156), (float)(-0.0744397065), (float)(-0.0129005460)};
  for (int i = 0; i < 4; ++i) m->insert[i] = ins[i];
  m->single[M] = (float)(0.3959924457);
  m->single[IX] = m->single[IY] = (float)(-0.4431756229);
  m->single[I2X] = m->single[I2Y] = (float)(-0.3488104904);
  const float m2m = (float)(2.5057567100), m2i = (float)(-1.2423961130), iext = (float)(1.8676346730), ichg = (float)(-6.9696754440);
  const float m2i2 = (float)(0.1970448791), i2ext = (float)(1.0140265830), i2chg = (float)(-7.3469687820);
  m->pair[M][M] = m2m;
  m->pair[M][IX] = m->pair[M][IY] = m->pair[IX][M] = m->pair[IY][M] = m2i;
  m->pair[IX][IX] = m->pair[IY][IY] = iext;
  m->pair[IX][IY] = m->pair[IY][IX] = ichg;
  m->pair[M][I2X] = m->pair[M][I2Y] = m->pair[I2X][M] = m->pair[I2Y][M] = m2i2;
  m->pair[I2X][I2X] = m->pair[I2Y][I2Y] = i2ext;
  m->pair[I2X][I2Y] = m->pair[I2Y][I2X] = i2chg;
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
  // non-blocking: the folding must not be drawn into the implicit synchronisation of null-stream copies
  if (hip_check(hipStreamCreateWithFlags(&c->fold_stream, hipStreamNonBlocking))) { (void)hipStreamDestroy(c->stream); delete c; return DAFS_HIP_ENODEV; }
  if (hip_check(hipStreamCreateWithFlags(&c->node_stream, hipStreamNonBlocking))) { (void)hipStreamDestroy(c->fold_stream); (void)hipStreamDestroy(c->stream); delete c; return DAFS_HIP_ENODEV; }
  int cus = 0;
  if (!hip_check(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device)) && cus > 0) c->num_cus = cus;
  *out = c;
  return DAFS_HIP_OK;
}

// ---- per-kernel device timings (bench.py "stages"; stage.h) ----
extern "C" int dafs_hip_stage_timing(dafs_hip_ctx* c, int enable) {
  if (!c) return DAFS_HIP_EINVAL;
  if (hip_check(hipSetDevice(c->device))) return DAFS_HIP_ENODEV;
  if (hip_check(hipDeviceSynchronize())) return DAFS_HIP_ELAUNCH;
  c->stages.clear();
  if (enable) stage_current() = &c->stages;
  else if (stage_current() == &c->stages) stage_current() = nullptr;
  return DAFS_HIP_OK;
}

extern "C" int dafs_hip_stage_report(dafs_hip_ctx* c, dafs_stage_time* out, uint32_t cap, uint32_t* n) {
  if (!c || !n || (cap && !out)) return DAFS_HIP_EINVAL;
  if (hip_check(hipSetDevice(c->device))) return DAFS_HIP_ENODEV;
  if (hip_check(hipDeviceSynchronize())) return DAFS_HIP_ELAUNCH;
  double ms[ST_COUNT] = {0};
  double longest[ST_COUNT] = {0};
  uint32_t cnt[ST_COUNT] = {0};
  for (const stage_recorder::rec& r : c->stages.recs) {
    float t = 0.0f;
    if (hip_check(hipEventElapsedTime(&t, r.a, r.b))) return DAFS_HIP_ELAUNCH;
    ms[r.id] += t;
    if (t > longest[r.id]) longest[r.id] = t;
    ++cnt[r.id];
  }
  uint32_t k = 0;
  for (int id = 0; id < ST_COUNT; ++id) {
    if (!cnt[id]) continue;
    if (k < cap) { out[k].kernel = kStageNames[id]; out[k].ms = ms[id]; out[k].longest_ms = longest[id]; out[k].launches = cnt[id]; }
    ++k;
  }
  *n = k;
  c->stages.clear();  // the next report starts from here
  return DAFS_HIP_OK;
}

extern "C" void dafs_hip_destroy(dafs_hip_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (stage_current() == &c->stages) stage_current() = nullptr;
  c->stages.release();
  c->free_all();
  (void)hipStreamDestroy(c->fold_stream);
  (void)hipStreamDestroy(c->node_stream);
  (void)hipStreamDestroy(c->stream);
  delete c;
}

extern "C" int dafs_hip_set_sequences(dafs_hip_ctx* c, uint32_t nseq, const char* const* seqs, const uint32_t* lens) {
  if (!c || !seqs || !lens || nseq == 0) return DAFS_HIP_EINVAL;
  if (hip_check(hipSetDevice(c->device))) return DAFS_HIP_ENODEV;
  c->len.assign(lens, lens + nseq);
  c->off.resize(nseq + 1);
  c->seq_rp_off.resize(nseq + 1);
  c->off[0] = 0;
  c->seq_rp_off[0] = 0;
  for (uint32_t i = 0; i < nseq; ++i) {
    if (lens[i] == 0) return DAFS_HIP_EINVAL;  // the reference reads seq[0] unconditionally (ProbabilisticModel.h:123-131)
    c->off[i + 1] = c->off[i] + lens[i];
    c->seq_rp_off[i + 1] = c->seq_rp_off[i] + lens[i] + 1;
  }
  c->seq.clear();
  c->seq.reserve(c->off[nseq]);
  std::vector<uint8_t> codes(c->off[nseq]);
  for (uint32_t i = 0; i < nseq; ++i) {
    c->seq.append(seqs[i], lens[i]);
    for (uint32_t k = 0; k < lens[i]; ++k) codes[c->off[i] + k] = dafs_hip_residue_code(seqs[i][k]);
  }
  int rc;
  if ((rc = c->codes.upload(codes.data(), codes.size(), c->stream))) return rc;
  if ((rc = c->d_len.upload(c->len.data(), nseq, c->stream))) return rc;
  if ((rc = c->d_seq_rp_off.upload(c->seq_rp_off.data(), nseq + 1, c->stream))) return rc;
  for (int k = 0; k < 2; ++k) { c->mp[k].valid = false; c->bp[k].valid = false; }
  c->cur_mp = c->cur_bp = 0;
  c->sim.clear();
  return DAFS_HIP_OK;
}

// make_brackets, reference src/nussinov.cpp:401-413 with brackets[0] = "()" (src/fold.cpp:57-58)
extern "C" void dafs_hip_make_brackets(uint32_t L, const uint32_t* ss, char* str) {
  memset(str, '.', L);
  str[L] = 0;
  for (uint32_t i = 0; i != L; ++i)
    if (ss[i] != DAFS_HIP_NONE) { str[i] = '('; str[ss[i]] = ')'; }
}
