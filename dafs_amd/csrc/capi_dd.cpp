// dafs_amd/csrc/capi_dd.cpp -- L1: the decoder plugins (Fold::Decoder / Align::Decoder,
// reference src/fold.h:47-60, src/align.h:57-65) and the fused per-node solver
// (DAFS::align_alignments + DAFS::solve_by_dd, reference src/dafs.cpp:896-981, 1006-1295).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>

#include "../../include/dafs_hip.h"
#include "ctx.h"
#include "dd.h"
#include "hip_util.h"

using namespace dafs;

namespace {

// bump allocator over one device buffer (256-byte aligned pieces)
struct carver {
  uint8_t* base = nullptr;
  size_t used = 0;
  template <class T>
  T* take(size_t n) {
    used = (used + 255) & ~(size_t)255;
    T* p = base ? (T*)(base + used) : nullptr;
    used += n * sizeof(T);
    return p;
  }
};

struct region { size_t off, bytes; int value; };

void carve_nuss(carver& cv, uint32_t L, nuss_ws& w);
// What only the folding DPs touch -- their work arrays (16 L^2 bytes each), the HBM copies of the traceback codes and the
// pair scores in the order the DPs read them (sweep order for the column-owning forms unless no launch uses them, by span
// for the span and workgroup forms; nd.lds_flags / nd.fold_fast say which) -- goes into the node's SECOND block, which is
// carved when the consensus-pair count is known: a node that leaves its foldings out (dafs_dd_params::skip_uncoupled_folds and
// no consensus pair) gets none of it, a third of its memory instead of all (27 GB -> 9 GB at the 27 000-column root of c5-random).
void carve_folding(carver& cv, dd_node& nd, bool force_wide) {
  const uint32_t L1 = nd.L1, L2 = nd.L2;
  const size_t XX = (size_t)L1 * L1, YY = (size_t)L2 * L2;
  carve_nuss(cv, L1, nd.wx);
  carve_nuss(cv, L2, nd.wy);
  nd.trk_x = nd.wx.tr; nd.trk_y = nd.wy.tr;  // the L*L uint32 tables double as bifurcation codes
  nd.trb_x = cv.take<uint8_t>(XX / 2 + L1 + 16); nd.trb_y = cv.take<uint8_t>(YY / 2 + L2 + 16);
  const bool span_only = (nd.lds_flags & 64u) != 0;
  const bool span_any = span_only || (nd.fold_fast & (16u | 32u)) != 0;
  nd.s_x = (dd_fold_cols(L1) <= DD_WFOLD && !force_wide && !span_only) ? cv.take<float>(((size_t)L1 + 63) * dd_fold_cols(L1) * 64) : nullptr;
  nd.s_y = (dd_fold_cols(L2) <= DD_WFOLD && !force_wide && !span_only) ? cv.take<float>(((size_t)L2 + 63) * dd_fold_cols(L2) * 64) : nullptr;
  nd.s_xs = (span_any || (nd.fold_fast & 64u)) ? cv.take<float>((size_t)L1 * ((L1 + 63) & ~63u) + 64) : nullptr;
  nd.s_ys = (span_any || (nd.fold_fast & 128u)) ? cv.take<float>((size_t)L2 * ((L2 + 63) & ~63u) + 64) : nullptr;
}

void carve_nuss(carver& cv, uint32_t L, nuss_ws& w) {
  const size_t LL = (size_t)L * L;
  w.dp = cv.take<float>(LL + 1);
  w.tr = cv.take<uint32_t>(LL + 1);
  w.ck = cv.take<uint32_t>(LL + 2 * (size_t)L + 16);  // doubles as the traceback stack
  w.cv = cv.take<float>(LL + 1);
  w.cc = cv.take<uint32_t>((size_t)L + 1);
}

}  // namespace

extern "C" int dafs_hip_nussinov_decode(dafs_hip_ctx* c, float th, float w, uint32_t L, const float* p, const float* q,
                                        uint32_t* ss, float* score) {
  if (!c || !p || !ss || L == 0) return DAFS_HIP_EINVAL;
  if (hip_check(hipSetDevice(c->device))) return DAFS_HIP_ENODEV;
  const size_t LL = (size_t)L * L;
  carver cv;
  for (int pass = 0; pass < 2; ++pass) {
    cv.used = 0;
    float* d_p = cv.take<float>(LL);
    float* d_q = q ? cv.take<float>(LL) : nullptr;
    nuss_ws ws;
    carve_nuss(cv, L, ws);
    uint32_t* d_ss = cv.take<uint32_t>(L);
    float* d_score = cv.take<float>(1);
    if (pass == 0) {
      int rc = c->work.reserve(cv.used + 256);
      if (rc) return rc;
      cv.base = c->work.ptr;
      continue;
    }
    if (hip_check(hipMemcpyAsync(d_p, p, LL * 4, hipMemcpyHostToDevice, c->stream))) return DAFS_HIP_ELAUNCH;
    if (q && hip_check(hipMemcpyAsync(d_q, q, LL * 4, hipMemcpyHostToDevice, c->stream))) return DAFS_HIP_ELAUNCH;
    int rc = nussinov_launch(L, d_p, d_q, w, th, ws, d_ss, d_score, c->stream);
    if (rc) return rc;
    if (hip_check(hipMemcpyAsync(ss, d_ss, (size_t)L * 4, hipMemcpyDeviceToHost, c->stream))) return DAFS_HIP_ELAUNCH;
    float s = 0;
    if (hip_check(hipMemcpyAsync(&s, d_score, 4, hipMemcpyDeviceToHost, c->stream))) return DAFS_HIP_ELAUNCH;
    if (hip_check(hipStreamSynchronize(c->stream))) return DAFS_HIP_ELAUNCH;
    if (score) *score = s;
  }
  return DAFS_HIP_OK;
}

static int nw_common(dafs_hip_ctx* c, float th, uint32_t L1, uint32_t L2, const float* p, const float* q, uint32_t* env,
                     int compute_env, int decode, uint32_t* al, float* score) {
  if (!c || !p || !env || L1 == 0 || L2 == 0) return DAFS_HIP_EINVAL;
  if (hip_check(hipSetDevice(c->device))) return DAFS_HIP_ENODEV;
  const size_t C = (size_t)L1 * L2, T = (size_t)(L1 + 1) * (L2 + 1);
  carver cv;
  for (int pass = 0; pass < 2; ++pass) {
    cv.used = 0;
    float* d_p = cv.take<float>(C);
    float* d_q = q ? cv.take<float>(C) : nullptr;
    float* d_dp = cv.take<float>(T + L1 + 2);
    uint8_t* d_tr = cv.take<uint8_t>(T);
    uint32_t* d_env = cv.take<uint32_t>(2 * ((size_t)L1 + 1));
    uint32_t* d_al = cv.take<uint32_t>((size_t)L1 + 2);
    float* d_score = cv.take<float>(1);
    if (pass == 0) {
      int rc = c->work.reserve(cv.used + 256);
      if (rc) return rc;
      cv.base = c->work.ptr;
      continue;
    }
    if (hip_check(hipMemcpyAsync(d_p, p, C * 4, hipMemcpyHostToDevice, c->stream))) return DAFS_HIP_ELAUNCH;
    if (q && hip_check(hipMemcpyAsync(d_q, q, C * 4, hipMemcpyHostToDevice, c->stream))) return DAFS_HIP_ELAUNCH;
    if (!compute_env && hip_check(hipMemcpyAsync(d_env, env, 2 * ((size_t)L1 + 1) * 4, hipMemcpyHostToDevice, c->stream))) return DAFS_HIP_ELAUNCH;
    int rc = nw_launch(L1, L2, d_p, d_q, th, d_env, compute_env, d_dp, d_tr, d_al, d_score, c->stream);
    if (rc) return rc;
    if (compute_env && hip_check(hipMemcpyAsync(env, d_env, 2 * ((size_t)L1 + 1) * 4, hipMemcpyDeviceToHost, c->stream))) return DAFS_HIP_ELAUNCH;
    float s = 0;
    if (decode) {
      if (hip_check(hipMemcpyAsync(al, d_al, (size_t)L1 * 4, hipMemcpyDeviceToHost, c->stream))) return DAFS_HIP_ELAUNCH;
      if (hip_check(hipMemcpyAsync(&s, d_score, 4, hipMemcpyDeviceToHost, c->stream))) return DAFS_HIP_ELAUNCH;
    }
    if (hip_check(hipStreamSynchronize(c->stream))) return DAFS_HIP_ELAUNCH;
    if (decode && score) *score = s;
  }
  return DAFS_HIP_OK;
}

extern "C" int dafs_hip_nw_envelope(dafs_hip_ctx* c, float th, uint32_t L1, uint32_t L2, const float* p, uint32_t* env) {
  return nw_common(c, th, L1, L2, p, nullptr, env, 1, 0, nullptr, nullptr);
}

extern "C" int dafs_hip_nw_decode(dafs_hip_ctx* c, float th, uint32_t L1, uint32_t L2, const float* p, const float* q,
                                  const uint32_t* env, uint32_t* al, float* score) {
  if (!al) return DAFS_HIP_EINVAL;
  return nw_common(c, th, L1, L2, p, q, (uint32_t*)env, 0, 1, al, score);
}

// The dense decoder classes (reference Nussinov, src/nussinov.cpp:32-204, and NeedlemanWunsch,
// src/needleman_wunsch.cpp:28-196; DAFS itself instantiates the sparse ones, src/dafs.cpp:1692,1759).
extern "C" int dafs_hip_nussinov_decode_dense(dafs_hip_ctx* c, float th, float w, uint32_t L, const float* p, const float* q,
                                              uint32_t* ss, float* score) {
  if (!c || !p || !ss || L == 0) return DAFS_HIP_EINVAL;
  if (hip_check(hipSetDevice(c->device))) return DAFS_HIP_ENODEV;
  const size_t LL = (size_t)L * L;
  carver cv;
  for (int pass = 0; pass < 2; ++pass) {
    cv.used = 0;
    float* d_p = cv.take<float>(LL);
    float* d_q = q ? cv.take<float>(LL) : nullptr;
    float* d_dp = cv.take<float>(LL + 1);
    uint32_t* d_tr = cv.take<uint32_t>(LL + 1);
    uint32_t* d_stack = cv.take<uint32_t>(4 * ((size_t)L + 4));
    uint32_t* d_ss = cv.take<uint32_t>(L);
    float* d_score = cv.take<float>(1);
    if (pass == 0) {
      int rc = c->work.reserve(cv.used + 256);
      if (rc) return rc;
      cv.base = c->work.ptr;
      continue;
    }
    if (hip_check(hipMemcpyAsync(d_p, p, LL * 4, hipMemcpyHostToDevice, c->stream))) return DAFS_HIP_ELAUNCH;
    if (q && hip_check(hipMemcpyAsync(d_q, q, LL * 4, hipMemcpyHostToDevice, c->stream))) return DAFS_HIP_ELAUNCH;
    int rc = nussinov_dense_launch(L, d_p, d_q, w, th, d_dp, d_tr, d_stack, d_ss, d_score, c->stream);
    if (rc) return rc;
    if (hip_check(hipMemcpyAsync(ss, d_ss, (size_t)L * 4, hipMemcpyDeviceToHost, c->stream))) return DAFS_HIP_ELAUNCH;
    float s = 0;
    if (hip_check(hipMemcpyAsync(&s, d_score, 4, hipMemcpyDeviceToHost, c->stream))) return DAFS_HIP_ELAUNCH;
    if (hip_check(hipStreamSynchronize(c->stream))) return DAFS_HIP_ELAUNCH;
    if (score) *score = s;
  }
  return DAFS_HIP_OK;
}

// NeedlemanWunsch::decode = the sparse decoder's DP with every cell inside the envelope
extern "C" int dafs_hip_nw_decode_dense(dafs_hip_ctx* c, float th, uint32_t L1, uint32_t L2, const float* p, const float* q,
                                        uint32_t* al, float* score) {
  if (!al || !L1 || !L2) return DAFS_HIP_EINVAL;
  std::vector<uint32_t> env(2 * ((size_t)L1 + 1));
  for (uint32_t i = 0; i <= L1; ++i) { env[2 * i] = i ? 1u : 0u; env[2 * i + 1] = L2; }
  return nw_common(c, th, L1, L2, p, q, env.data(), 0, 1, al, score);
}

// ---------------------------------------------------------------------------------------------
// per-node solver
// ---------------------------------------------------------------------------------------------
namespace {

struct geom {  // host-side geometry of one child alignment
  std::vector<uint32_t> rank, idx, idxoff;
};

int make_geom(const dafs_hip_ctx* c, uint32_t n, uint32_t L, const uint32_t* seq, const uint8_t* mask, geom& g) {
  g.rank.assign((size_t)n * L, DAFS_HIP_NONE);
  g.idxoff.resize(n);
  g.idx.clear();
  for (uint32_t r = 0; r < n; ++r) {
    if (seq[r] >= c->len.size()) return DAFS_HIP_EINVAL;
    g.idxoff[r] = (uint32_t)g.idx.size();
    uint32_t k = 0;
    for (uint32_t i = 0; i < L; ++i)
      if (mask[(size_t)r * L + i]) { g.rank[(size_t)r * L + i] = k++; g.idx.push_back(i); }
    if (k != c->len[seq[r]]) return DAFS_HIP_EINVAL;  // the mask must place every residue
  }
  return DAFS_HIP_OK;
}

}  // namespace

extern "C" void dafs_hip_dd_default_params(dafs_dd_params* p) {
  if (!p) return;
  p->w = 4.0f; p->eta0 = 0.5f; p->th_a = 0.01f; p->th_s = 0.2f; p->t_max = 600; p->force_iters = 0; p->skip_uncoupled_folds = 0;  // dafs.cpp:1612-1640
}

namespace {

dd_params device_params(const dafs_dd_params* prm) {
  dd_params dp;
  dp.w = prm->w; dp.eta0 = prm->eta0; dp.th_a = prm->th_a; dp.th_s = prm->th_s; dp.t_max = prm->t_max; dp.force_iters = prm->force_iters;
  dp.stamps = getenv("DAFS_HIP_DD_STAMPS") ? 1 : 0;
  dp.skip_xy = prm->skip_uncoupled_folds ? 1 : 0;
  dp.debug_lose_folders = getenv("DAFS_HIP_DD_LOSE_FOLDERS") ? 1 : 0;
  {
    const char* e = getenv("DAFS_HIP_DD_SPAN_MW");
    dp.span_one_wave = (e && atoi(e) == 0) ? 1 : 0;
  }
  dp.slice = 0;
  dp.budget = 0; dp.t_ref = nullptr; dp.t_ref_write = 0;
  return dp;
}

// Builds nnodes resident nodes (appended to c->dd_open): geometry upload, profile averages, sparse lists and
// consensus constraints (DAFS::align_alignments up to the solve_by_dd call, dafs.cpp:896-960).
// A lane = a stream with the node-descriptor and per-node-word buffers its launches use.  Lane 0 is the context's main
// stream; dafs_hip_nodes_round sets up and starts new nodes on lane 1 while the open ones advance on lane 0.
struct dd_lane { hipStream_t st; dev_buf<dd_node>* d_nodes; dev_buf<uint32_t>* d_paused; int id; };
dd_lane lane_of(dafs_hip_ctx* c, int k) { return k == 0 ? dd_lane{c->stream, &c->d_nodes, &c->d_paused, 0} : dd_lane{c->node_stream, &c->d_nodes2, &c->d_paused2, 1}; }

// What a node's launch would read, checked on the host before anything is enqueued.  Every form of the folding DPs
// reads one of the two score copies (s_x / s_y in sweep order for the column-owning register forms, s_xs / s_ys by span
// for the span form), and nodes_open leaves out the copies no form of the node can use.  A plan that selects a form
// whose copy is absent would make the kernel use a null base + cell offset as an address: that was the memory-access
// fault of round 2 ("the fault at 16 x ~1100 columns": the multiplier updates wrote s_x[skew(i, j)] of
// nodes beyond 1024 columns, whose s_x had just become optional).  The kernel's writes are guarded now; this check
// turns any future mismatch between the carving and the form selection into DAFS_HIP_ELAUNCH instead of a fault.
// folds: this launch runs the node's folding DPs (the kernel's fold_on: not (skip_uncoupled_folds and no consensus pair))
int plan_check(const dd_node& nd, bool split, bool folds) {
  auto bad = [](const char* what) {
    fprintf(stderr, "dafs_hip: node plan refused: %s\n", what);
    return DAFS_HIP_ELAUNCH;
  };
  const void* always[] = {nd.seq1, nd.seq2, nd.rank1, nd.rank2, nd.idx1, nd.idx2, nd.idxoff1, nd.idxoff2, nd.p_x, nd.p_y, nd.p_z, nd.q_x, nd.q_y, nd.q_z,
                          nd.nw_edge, nd.tr_z, nd.pz_s, nd.qz_s, nd.env, nd.env4, nd.xmap, nd.ymap, nd.zmap, nd.px_ptr, nd.px_j,
                          nd.py_ptr, nd.py_l, nd.pz_ptr, nd.pz_k, nd.cz_ptr, nd.cz_k, nd.cx_flag, nd.cy_flag, nd.cz_flag, nd.cbp_cnt, nd.cbp, nd.sw,
                          nd.tx, nd.ty, nd.tz, nd.x, nd.y, nd.z, nd.score, nd.info, nd.fstate, nd.sync};
  for (const void* q : always)
    if (!q) return bad("a null array in the node descriptor");
  if (!nd.L1 || !nd.L2 || !nd.n1 || !nd.n2) return bad("empty child alignment");
  if (!folds) return DAFS_HIP_OK;  // nothing below is touched
  const void* folding[] = {nd.wx.dp, nd.wx.tr, nd.wx.ck, nd.wx.cv, nd.wx.cc, nd.wy.dp, nd.wy.tr, nd.wy.ck, nd.wy.cv, nd.wy.cc, nd.trb_x, nd.trb_y, nd.trk_x, nd.trk_y};
  for (const void* q : folding)
    if (!q) return bad("a node that folds without its folding arrays (opened with skip_uncoupled_folds, advanced without?)");
  const bool regx = dd_fold_cols(nd.L1) <= DD_WFOLD, regy = dd_fold_cols(nd.L2) <= DD_WFOLD;
  if (!split) {
    const uint32_t f = nd.lds_flags;
    if ((f & 64u) && (!nd.s_xs || !nd.s_ys)) return bad("span form without the by-span score copies");
    if ((f & 64u) && (nd.L1 > DD_SPAN_LMAX || nd.L2 > DD_SPAN_LMAX)) return bad("span form beyond its width");
    if (!(f & 64u) && (f & (2u | 8u)) && regx && !nd.s_x) return bad("register form of the x folding without its sweep-order scores");
    if (!(f & 64u) && (f & (4u | 8u)) && regy && !nd.s_y) return bad("register form of the y folding without its sweep-order scores");
  } else {
    if (nd.lds_flags & ~1u) return bad("a split leader keeps the alignment DP only");
    for (int r = 0; r < 2; ++r) {
      const uint32_t L = r ? nd.L2 : nd.L1;
      const bool reg = r ? regy : regx;
      const float* sweep = r ? nd.s_y : nd.s_x;
      const float* byspan = r ? nd.s_ys : nd.s_xs;
      if (nd.fold_fast & (16u << r)) {
        if (!byspan) return bad("span-form folder without the by-span score copy");
        if (L > DD_SPAN_LMAX) return bad("span-form folder beyond its width");
      } else if ((nd.fold_fast & (5u << r)) && reg && !sweep) return bad("register-form folder without its sweep-order scores");
      if (nd.fold_fast & (64u << r)) {
        if (!byspan) return bad("workgroup-form folder without the by-span score copy");
      }
    }
  }
  return DAFS_HIP_OK;
}

struct open_blocks {  // device blocks of the nodes an open call has carved so far (given back when the call fails)
  std::vector<uint8_t*> blk0, blk1;
  std::vector<size_t> bytes0, bytes1;
};

// DAFS_HIP_DD_FAIL_OPEN=k (tests): nodes_open fails with DAFS_HIP_ELAUNCH at its k-th stage (1 after the blocks and their
// fills are queued, 2 after the lists, 3 after the second blocks, 4 after everything) -- the late-failure path
bool fail_injected(int stage) {
  const char* e = getenv("DAFS_HIP_DD_FAIL_OPEN");
  return e && atoi(e) == stage;
}

int nodes_open_impl(dafs_hip_ctx* c, const dd_lane& ln, uint32_t nnodes, const dafs_node_input* in, const dd_params& dp, open_blocks& ob) {
  const mp_store& mps = c->mp[c->cur_mp];
  const bp_store& bps = c->bp[c->cur_bp];
  const uint32_t nseq = (uint32_t)c->len.size();
  if (!mps.valid || !bps.valid || mps.n_tasks != (uint64_t)nseq * (nseq - 1) / 2) return DAFS_HIP_EINVAL;

  // ---- host geometry ----
  std::vector<geom> g1(nnodes), g2(nnodes);
  for (uint32_t b = 0; b < nnodes; ++b) {
    const dafs_node_input& ni = in[b];
    if (!ni.n1 || !ni.n2 || !ni.len1 || !ni.len2 || !ni.seq1 || !ni.seq2 || !ni.mask1 || !ni.mask2) return DAFS_HIP_EINVAL;
    int rc;
    if ((rc = make_geom(c, ni.n1, ni.len1, ni.seq1, ni.mask1, g1[b]))) return rc;
    if ((rc = make_geom(c, ni.n2, ni.len2, ni.seq2, ni.mask2, g2[b]))) return rc;
  }

  // DAFS_HIP_DD_WIDE=1 (tests): every node takes the forms of alignments too wide for the on-chip placements -- foldings
  // span-ordered on HBM tables without sweep-order copies, the alignment DP in panels of 64 columns with its codes in
  // HBM slots, row pointers searched in HBM, one averaging row per workgroup
  const char* wide_env = getenv("DAFS_HIP_DD_WIDE");
  const bool force_wide = wide_env && atoi(wide_env) != 0;
  // ---- carve each node's block (two passes: size, then pointers) ----
  std::vector<dd_node> nodes(nnodes);
  std::vector<std::vector<uint8_t>> heads(nnodes);  // upload staging, alive until the first synchronisation below
  std::vector<size_t> lds(nnodes, 0), split_lds(nnodes, 0);
  ob.blk0.assign(nnodes, nullptr); ob.blk1.assign(nnodes, nullptr);
  ob.bytes0.assign(nnodes, 0); ob.bytes1.assign(nnodes, 0);
  std::vector<uint8_t*>&blk0 = ob.blk0, &blk1 = ob.blk1;
  std::vector<size_t>&blk0_bytes = ob.bytes0, &blk1_bytes = ob.bytes1;
  for (uint32_t b = 0; b < nnodes; ++b) {
    const dafs_node_input& ni = in[b];
    dd_node& nd = nodes[b];
    carver cv;
    std::vector<region> fills;
    for (int pass = 0; pass < 2; ++pass) {
      cv.used = 0;
      fills.clear();
      memset(&nd, 0, sizeof nd);
      const uint32_t L1 = ni.len1, L2 = ni.len2;
      nd.n1 = ni.n1; nd.n2 = ni.n2; nd.L1 = L1; nd.L2 = L2;
      const size_t XX = (size_t)L1 * L1, YY = (size_t)L2 * L2, ZZ = (size_t)L1 * L2;
      nd.seq1 = cv.take<uint32_t>(ni.n1); nd.seq2 = cv.take<uint32_t>(ni.n2);
      nd.rank1 = cv.take<uint32_t>((size_t)ni.n1 * L1); nd.rank2 = cv.take<uint32_t>((size_t)ni.n2 * L2);
      nd.idx1 = cv.take<uint32_t>(g1[b].idx.size() + 1); nd.idx2 = cv.take<uint32_t>(g2[b].idx.size() + 1);
      nd.idxoff1 = cv.take<uint32_t>(ni.n1); nd.idxoff2 = cv.take<uint32_t>(ni.n2);
      // zero-filled block: posteriors, multipliers, flags
      const size_t z0 = (cv.used + 255) & ~(size_t)255;
      nd.p_x = cv.take<float>(XX); nd.p_y = cv.take<float>(YY); nd.p_z = cv.take<float>(ZZ);
      nd.q_x = cv.take<float>(XX); nd.q_y = cv.take<float>(YY); nd.q_z = cv.take<float>(ZZ);
      nd.cz_flag = cv.take<uint8_t>(ZZ);
      nd.cx_flag = cv.take<uint8_t>(XX / 2 + 2); nd.cy_flag = cv.take<uint8_t>(YY / 2 + 2);
      nd.sync = cv.take<uint32_t>(8);
      fills.push_back({z0, cv.used - z0, 0});
      // -1-filled block: dense id maps
      const size_t m0 = (cv.used + 255) & ~(size_t)255;
      nd.xmap = cv.take<int32_t>(XX); nd.ymap = cv.take<int32_t>(YY); nd.zmap = cv.take<int32_t>(ZZ);
      fills.push_back({m0, cv.used - m0, 0xFF});
      // (the folding DPs' work arrays, codes and score copies are carved into the node's second block, once the
      // consensus-pair count says whether this node folds at all: carve_folding below)
      // the alignment DP: columns per lane, and with them the panels of second alignments beyond 64 nw_w - 1 columns
      nd.nw_w = force_wide ? 1u : dd_nw_cols(L2);
      const size_t nw_panels = dd_nw_panels(L2, nd.nw_w);
      nd.nw_edge = cv.take<float>(2 * ((size_t)L1 + 2));
      nd.tr_z = cv.take<uint8_t>(nw_panels * (L1 + 1) * 512);  // a 64-bit slot per (panel, row, lane)
      // sweep-order inputs of the alignment DP: steps x columns per lane x 64 lanes, panel by panel
      nd.pz_s = cv.take<float>(nw_panels * ((size_t)L1 + 63) * nd.nw_w * 64); nd.qz_s = cv.take<float>(nw_panels * ((size_t)L1 + 63) * nd.nw_w * 64);
      {  // LDS plan (mirrors the carving at the top of k_dd_solve / dd_folder)
        auto nib = [](uint32_t L) { return ((size_t)L * (L + 1) / 2 + 7) / 8; };           // packed traceback codes, words
        // a fast folding DP: codes, the rows in flight (one per active lane), DD_CAP split rows per column
        auto fast = [&](uint32_t L) { return (nib(L) + dd_ring_words(L) + (size_t)DD_CAP * L) * 4; };
        const size_t need_z = (size_t)dd_nwtab_words(L1, L2) * 4;                           // packed alignment traceback
        const uint32_t Lm = std::max(L1, L2);
        const size_t shared = (std::max(nib(L1), nib(L2)) + std::max(dd_ring_words(L1), dd_ring_words(L2)) + (size_t)DD_CAP * Lm) * 4;
        const size_t shared_g = (std::max(dd_ring_words(L1), dd_ring_words(L2)) + (size_t)DD_CAP * Lm) * 4;
        auto fast_g = [&](uint32_t L) { return ((size_t)dd_ring_words(L) + (size_t)DD_CAP * L) * 4; };  // traceback codes in HBM
        auto wide_ok = [](uint32_t L, uint32_t cols) { return dd_fold_cols(L) <= cols; };  // a register form exists for this width
        size_t used = 0;
        nd.lds_flags = 0;
        // span form of both foldings side by side (whole dp triangles on chip: up to ~170 + 170 columns); DAFS_HIP_DD_SPAN=0
        // keeps the column-owning forms (tests run both)
        const char* span_env = getenv("DAFS_HIP_DD_SPAN");
        const bool span_allowed = !(span_env && atoi(span_env) == 0);
        const size_t span_xy = ((size_t)dd_span_words(L1) + dd_span_words(L2)) * 4 + 16;
        if (force_wide) {}
        else if (span_allowed && L1 <= DD_SPAN_LMAX && L2 <= DD_SPAN_LMAX && used + span_xy + need_z <= kDdLdsBudget) { used += span_xy; nd.lds_flags |= 64u; }  // only with the alignment traceback on chip too
        else if (used + fast(L1) + fast(L2) <= kDdLdsBudget) { used += fast(L1) + fast(L2); nd.lds_flags |= 2u | 4u; }  // x and y side by side
        else if (wide_ok(L1, DD_WREG) && wide_ok(L2, DD_WREG) && used + shared <= kDdLdsBudget) { used += shared; nd.lds_flags |= 8u; }      // one region, x then y
        else if (wide_ok(L1, DD_WFOLD) && wide_ok(L2, DD_WFOLD) && used + shared_g <= kDdLdsBudget) { used += shared_g; nd.lds_flags |= 8u | 16u; }  // the same, codes in HBM
        // else: foldings without a register form run span-ordered on HBM tables and need no LDS
        // DAFS_HIP_DD_NWG=1 (tests): no alignment codes in LDS, so that every node takes the register form with its codes in HBM slots
        if (used + need_z <= kDdLdsBudget && !force_wide && !getenv("DAFS_HIP_DD_NWG") && nd.nw_w <= DD_WNW && L2 < 64u * nd.nw_w) { used += need_z; nd.lds_flags |= 1u; }
        lds[b] = used;
        // split plan: each folding DP on a workgroup of its own.  Worth it when the two do not run side by side
        // in one workgroup; the leader then keeps only the alignment DP (its LDS need is covered by `used`).
        nd.split = 0; nd.fold_fast = 0;
        split_lds[b] = 0;
        // Nodes whose foldings cannot take the span form side by side, but can on a workgroup of their own, are worth
        // splitting even when the column-owning forms fit side by side: the span form is about twice as fast.
        // ... and since round 3 also when they do fit side by side, as soon as a folding has more than one row slot: its
        // folder shares the slots out to its wavefronts (nuss_span_mw: a span costs one slot step + a barrier instead of
        // one slot step per live slot), which the node's own workgroup -- one wavefront per subproblem -- cannot do.
        const char* mw_env = getenv("DAFS_HIP_DD_SPAN_MW");
        const bool mw_allowed = !(mw_env && atoi(mw_env) == 0);
        const char* wg_env = getenv("DAFS_HIP_DD_WG");
        const bool wg_allowed = !(wg_env && atoi(wg_env) == 0);
        const bool wg_force = wg_env && atoi(wg_env) == 2;  // tests: every folder takes the workgroup form, whatever its width
        const bool span_folders = span_allowed && !force_wide && L1 <= DD_SPAN_LMAX && L2 <= DD_SPAN_LMAX &&
                                  (size_t)dd_span_words(std::max(L1, L2)) * 4 + 16 <= kDdLdsBudget &&
                                  (!(nd.lds_flags & 64u) || (mw_allowed && std::max(L1, L2) > 64));
        if ((!(nd.lds_flags & (2u | 64u)) || span_folders || wg_force) && !force_wide) {
          size_t worst = 0;
          const uint32_t Ls[2] = {L1, L2};
          for (int r = 0; r < 2; ++r) {
            const uint32_t L = Ls[r];
            const bool no_reg = !wide_ok(L, DD_WFOLD) || wg_force;
            if (span_folders && !wg_force) { nd.fold_fast |= 16u << r; worst = std::max(worst, (size_t)dd_span_words(L) * 4 + 16); }
            else if (!no_reg && wide_ok(L, DD_WREG) && fast(L) <= kDdLdsBudget) { nd.fold_fast |= 1u << r; worst = std::max(worst, fast(L)); }
            else if (!no_reg && wide_ok(L, DD_WFOLD) && fast_g(L) <= kDdLdsBudget) { nd.fold_fast |= 4u << r; worst = std::max(worst, fast_g(L)); }
            else if (no_reg && wg_allowed) {
              // no register form: the workgroup form with as many candidates per column on chip as fit (dd_wg_words); beyond
              // ~10 000 columns not even its rolling rows fit and the span-ordered form on HBM tables remains
              for (uint32_t K : {4u, 2u, 0u})
                if ((size_t)dd_wg_words(L, K) * 4 + 16 <= kDdLdsBudget) {
                  nd.fold_fast |= (64u << r) | (K << (8 + 4 * r));
                  worst = std::max(worst, (size_t)dd_wg_words(L, K) * 4 + 16);
                  break;
                }
            }
            // else span-ordered on HBM tables: no LDS
          }
          // also worth it when a folding has no register form at all: its folder runs the span-ordered form on a
          // whole workgroup, far ahead of the HBM-table wave form the leader would run for it
          if (nd.fold_fast || !wide_ok(L1, DD_WFOLD) || !wide_ok(L2, DD_WFOLD)) split_lds[b] = std::max(worst, used);
        }
      }
      nd.env = cv.take<uint32_t>(2 * ((size_t)L1 + 1));
      nd.env4 = cv.take<uint32_t>(2 * ((size_t)L1 + 130));
      nd.px_ptr = cv.take<uint32_t>((size_t)L1 + 2); nd.px_j = cv.take<uint32_t>(XX / 2 + 2);
      nd.py_ptr = cv.take<uint32_t>((size_t)L2 + 2); nd.py_l = cv.take<uint32_t>(YY / 2 + 2);
      nd.pz_ptr = cv.take<uint32_t>((size_t)L1 + 2); nd.pz_k = cv.take<uint32_t>(ZZ + 1);
      nd.cz_ptr = cv.take<uint32_t>((size_t)L1 + 2); nd.cz_k = cv.take<uint32_t>(ZZ + 1);
      nd.cbp_cnt = cv.take<uint32_t>(XX / 2 + 2);
      nd.tx = cv.take<int32_t>(XX / 2 + 2); nd.ty = cv.take<int32_t>(YY / 2 + 2); nd.tz = cv.take<int32_t>(ZZ + 1);
      nd.x = cv.take<uint32_t>((size_t)L1 + 2); nd.y = cv.take<uint32_t>((size_t)L2 + 2); nd.z = cv.take<uint32_t>((size_t)L1 + 2);
      nd.score = cv.take<float>(1); nd.info = cv.take<uint32_t>(16); nd.fstate = cv.take<float>(4);
      if (pass == 0) {
        cv.base = c->dd_alloc(cv.used + 256);
        if (!cv.base) return DAFS_HIP_ENOMEM;
        blk0[b] = cv.base; blk0_bytes[b] = cv.used + 256;
      }
    }
    for (const region& r : fills)
      if (hip_check(hipMemsetAsync(cv.base + r.off, r.value, r.bytes, ln.st))) return DAFS_HIP_ELAUNCH;
    // the geometry arrays were carved first and back to back: one upload of the head of the block brings them all
    {
      const size_t head = (size_t)((const uint8_t*)(nd.idxoff2 + ni.n2) - cv.base);
      std::vector<uint8_t>& blob = heads[b];
      blob.assign(head, 0);
      auto put = [&](const void* dst, const void* src, size_t bytes) {
        if (bytes) memcpy(blob.data() + ((const uint8_t*)dst - cv.base), src, bytes);
      };
      put(nd.seq1, ni.seq1, (size_t)ni.n1 * 4); put(nd.seq2, ni.seq2, (size_t)ni.n2 * 4);
      put(nd.rank1, g1[b].rank.data(), g1[b].rank.size() * 4); put(nd.rank2, g2[b].rank.data(), g2[b].rank.size() * 4);
      put(nd.idx1, g1[b].idx.data(), g1[b].idx.size() * 4); put(nd.idx2, g2[b].idx.data(), g2[b].idx.size() * 4);
      put(nd.idxoff1, g1[b].idxoff.data(), (size_t)ni.n1 * 4); put(nd.idxoff2, g2[b].idxoff.data(), (size_t)ni.n2 * 4);
      if (hip_check(hipMemcpyAsync(cv.base, blob.data(), head, hipMemcpyHostToDevice, ln.st))) return DAFS_HIP_ELAUNCH;
    }
  }
  int rc;
  if (fail_injected(1)) return DAFS_HIP_ELAUNCH;
  if ((rc = ln.d_nodes->upload(nodes.data(), nnodes, ln.st))) return rc;  // synchronises: host vectors stay valid until here
  const mp_store_dev mpv = mps.view(c->d_len.ptr, nseq);
  const bp_store_dev bpv = bps.view();
  uint32_t max_len = 0;
  for (uint32_t b = 0; b < nnodes; ++b) max_len = std::max(max_len, std::max(in[b].len1, in[b].len2));
  // few nodes with hundreds of source rows per row of p_z (the top of the guide tree): a workgroup per p_z row
  uint64_t srcs = 0;
  for (uint32_t b = 0; b < nnodes; ++b) srcs = std::max<uint64_t>(srcs, (uint64_t)in[b].n1 * in[b].n2);
  const int coop = (nnodes <= 4 && srcs >= 512 && !getenv("DAFS_HIP_AVG_COOP0")) ? 1 : 0;
  if ((rc = dd_avg_launch(ln.d_nodes->ptr, nnodes, max_len, mpv, bpv, force_wide ? 1 : 0, coop, ln.st))) return rc;
  for (uint32_t b = 0; b < nnodes; ++b) {  // base-pairing matrices supplied by the caller (--bp-update) replace the averages
    const size_t XX = (size_t)in[b].len1 * in[b].len1, YY = (size_t)in[b].len2 * in[b].len2;
    if (in[b].p_x && hip_check(hipMemcpyAsync(nodes[b].p_x, in[b].p_x, XX * 4, hipMemcpyHostToDevice, ln.st))) return DAFS_HIP_ELAUNCH;
    if (in[b].p_y && hip_check(hipMemcpyAsync(nodes[b].p_y, in[b].p_y, YY * 4, hipMemcpyHostToDevice, ln.st))) return DAFS_HIP_ELAUNCH;
  }
  if ((rc = ln.d_paused->reserve(nnodes))) return rc;  // doubles as the landing place of the per-node counts
  if ((rc = dd_lists_launch(ln.d_nodes->ptr, nnodes, force_wide ? 0 : max_len, dp, ln.d_paused->ptr, ln.st))) return rc;
  // ---- consensus base-pair counts -> each node's second block ----
  std::vector<uint32_t> counts(nnodes);
  if (hip_check(hipMemcpyAsync(counts.data(), ln.d_paused->ptr, (size_t)nnodes * 4, hipMemcpyDeviceToHost, ln.st))) return DAFS_HIP_ELAUNCH;
  if (hip_check(hipStreamSynchronize(ln.st))) return DAFS_HIP_ELAUNCH;
  if (fail_injected(2)) return DAFS_HIP_ELAUNCH;
  for (uint32_t b = 0; b < nnodes; ++b) {
    const uint32_t ncbp = counts[b];
    carver cb;
    for (int pass = 0; pass < 2; ++pass) {
      cb.used = 0;
      nodes[b].ncbp_cap = ncbp;
      nodes[b].cbp = cb.take<uint32_t>((size_t)8 * ncbp + 8);
      nodes[b].sw = cb.take<float>((size_t)ncbp + 1);
      if (!(dp.skip_xy && ncbp == 0)) carve_folding(cb, nodes[b], force_wide);  // the kernel's fold_on
      if (pass == 0) {
        cb.base = c->dd_alloc(cb.used + 256);
        if (!cb.base) return DAFS_HIP_ENOMEM;
        blk1[b] = cb.base; blk1_bytes[b] = cb.used + 256;
      }
    }
  }
  if (fail_injected(3)) return DAFS_HIP_ELAUNCH;
  for (uint32_t b = 0; b < nnodes; ++b)  // both placements a launch may choose for this node, before anything runs on it
    if ((rc = plan_check(nodes[b], false, !(dp.skip_xy && counts[b] == 0)))) return rc;
  if ((rc = ln.d_nodes->upload(nodes.data(), nnodes, ln.st))) return rc;
  if ((rc = dd_cbp_fill_launch(ln.d_nodes->ptr, nnodes, force_wide ? 0 : max_len, dp, ln.st))) return rc;
  if (fail_injected(4)) return DAFS_HIP_ELAUNCH;
  for (uint32_t b = 0; b < nnodes; ++b) {
    dafs_hip_ctx::dd_open_node on;
    on.nd = nodes[b]; on.lds = lds[b]; on.split_lds = split_lds[b];
    on.blk[0] = blk0[b]; on.blk[1] = blk1[b]; on.blk_bytes[0] = blk0_bytes[b]; on.blk_bytes[1] = blk1_bytes[b];
    c->dd_open.push_back(on);
  }
  return DAFS_HIP_OK;
}

// A failed open leaves nothing behind: the fills, uploads and set-up kernels it has queued on the lane's stream may still
// write into the blocks it carved, so the stream is drained before they go back to the free list (the next dd_alloc,
// possibly for the other lane, may hand them out at once), and the nodes it may have appended are dropped.
int nodes_open(dafs_hip_ctx* c, const dd_lane& ln, uint32_t nnodes, const dafs_node_input* in, const dd_params& dp) {
  const size_t first = c->dd_open.size();
  open_blocks ob;
  const int rc = nodes_open_impl(c, ln, nnodes, in, dp, ob);
  if (rc) {
    (void)hipStreamSynchronize(ln.st);
    c->dd_open.resize(first);
    for (size_t b = 0; b < ob.blk0.size(); ++b) { c->dd_free(ob.blk0[b], ob.bytes0[b]); c->dd_free(ob.blk1[b], ob.bytes1[b]); }
  }
  return rc;
}

// One launch of the subgradient loop over the given resident nodes; finished[k] tells which of them are done.
// In two halves, so that launches on two lanes can be in flight together: advance_launch enqueues the kernel and the
// copy of the per-node words, advance_collect waits for them.
struct advance_state {
  std::vector<uint32_t> who, handles, off;
  const uint32_t* paused = nullptr;
  const uint32_t* packed = nullptr;
  bool launched = false;  // the solver may be running: collect must wait for the lane
  bool complete = false;  // ... and the per-node words are on their way to the landing place
};

int advance_launch(dafs_hip_ctx* c, const dd_lane& ln, uint32_t n, const uint32_t* handles, dd_params dp, uint32_t max_iterations, uint8_t* finished,
                   advance_state& stt) {
  std::vector<dd_node> nodes;
  std::vector<uint32_t>& who = stt.who;
  who.clear();
  stt.handles.assign(handles, handles + n);
  stt.launched = false;
  stt.complete = false;
  size_t lds_max = 0;
  for (uint32_t k = 0; k < n; ++k) {
    if (handles[k] >= c->dd_open.size()) return DAFS_HIP_EINVAL;
    dafs_hip_ctx::dd_open_node& on = c->dd_open[handles[k]];
    if (finished) finished[k] = on.finished ? 1 : 0;
    if (on.finished) continue;
    nodes.push_back(on.nd);
    who.push_back(k);
  }
  if (nodes.empty()) return DAFS_HIP_OK;
  // split mode (three workgroups per node) when the launch is small enough for all of them to be on the
  // machine at once and some node profits; DAFS_HIP_DD_SPLIT=0 turns it off
  const char* split_env = getenv("DAFS_HIP_DD_SPLIT");
  const bool split_allowed = !(split_env && atoi(split_env) == 0);
  bool split = false;
  // three workgroups per node, one per CU (their LDS does not leave room for a second): all of them must fit the device,
  // next to the workgroups of the other lane's launch when that one is still in flight (dafs_hip_nodes_round)
  const uint32_t other_wgs = c->dd_wgs_in_flight[ln.id ^ 1];
  if (split_allowed && (int)(nodes.size() * 3 + other_wgs) <= c->num_cus - 16)
    for (size_t b = 0; b < nodes.size(); ++b) split = split || (c->dd_open[handles[who[b]]].split_lds != 0 && !c->dd_open[handles[who[b]]].no_split);
  for (size_t b = 0; b < nodes.size(); ++b) {
    const dafs_hip_ctx::dd_open_node& on = c->dd_open[handles[who[b]]];
    if (split && on.split_lds && !on.no_split) {
      nodes[b].split = 1;
      nodes[b].lds_flags &= 1u;  // the leader keeps the alignment DP only
      lds_max = std::max(lds_max, on.split_lds);
      if (hip_check(hipMemsetAsync(nodes[b].sync, 0, 4, ln.st))) return DAFS_HIP_ELAUNCH;  // clear the exit mark of the last launch
    } else {
      nodes[b].split = 0;
      lds_max = std::max(lds_max, on.lds);
    }
  }
  dp.slice = max_iterations;
  int rc;
  for (size_t b = 0; b < nodes.size(); ++b)  // the form each node takes in THIS launch against what its block holds
    if ((rc = plan_check(nodes[b], nodes[b].split != 0, !(dp.skip_xy && nodes[b].ncbp_cap == 0)))) return rc;
  if ((rc = ln.d_nodes->upload(nodes.data(), nodes.size(), ln.st))) return rc;
  if ((rc = ln.d_paused->reserve(nodes.size()))) return rc;
  for (size_t b = 0; b < nodes.size(); ++b) c->dd_open[handles[who[b]]].in_flight = true;
  c->dd_wgs_in_flight[ln.id] = (uint32_t)nodes.size() * (split ? 3u : 1u);
  stt.launched = true;  // from here on advance_collect has something to wait for, whatever fails below
  if ((rc = dd_solve_launch(ln.d_nodes->ptr, (uint32_t)nodes.size(), dp, lds_max, split, ln.d_paused->ptr, ln.st))) return rc;
  // the result words of every node of the launch come along (one packed copy): a node that finishes here needs no
  // copy and no synchronisation of its own in dafs_hip_nodes_result
  stt.off.assign(nodes.size() + 1, 0);
  for (size_t b = 0; b < nodes.size(); ++b) stt.off[b + 1] = stt.off[b] + (uint32_t)((nodes[b].info + 16) - nodes[b].x);
  const size_t total = stt.off[nodes.size()];
  dev_buf<uint32_t>& d_off = c->d_pack_off[ln.id];
  dev_buf<uint32_t>& d_pack = c->d_pack[ln.id];
  if ((rc = d_off.reserve(nodes.size()))) return rc;
  if ((rc = d_pack.reserve(total))) return rc;
  uint32_t* landing = c->pinned_words(ln.id, 2 * nodes.size() + total);
  if (!landing) return DAFS_HIP_ENOMEM;
  memcpy(landing + nodes.size(), stt.off.data(), nodes.size() * 4);  // staged in pinned memory: the upload stays asynchronous
  if (hip_check(hipMemcpyAsync(d_off.ptr, landing + nodes.size(), nodes.size() * 4, hipMemcpyHostToDevice, ln.st))) return DAFS_HIP_ELAUNCH;
  if ((rc = dd_pack_launch(ln.d_nodes->ptr, (uint32_t)nodes.size(), d_off.ptr, d_pack.ptr, ln.st))) return rc;
  stt.paused = landing;
  stt.packed = landing + 2 * nodes.size();
  if (hip_check(hipMemcpyAsync(landing, ln.d_paused->ptr, nodes.size() * 4, hipMemcpyDeviceToHost, ln.st))) return DAFS_HIP_ELAUNCH;
  if (total && hip_check(hipMemcpyAsync(landing + 2 * nodes.size(), d_pack.ptr, total * 4, hipMemcpyDeviceToHost, ln.st))) return DAFS_HIP_ELAUNCH;
  stt.complete = true;
  return DAFS_HIP_OK;
}

int advance_collect(dafs_hip_ctx* c, const dd_lane& ln, advance_state& stt, uint8_t* finished) {
  if (!stt.launched) return DAFS_HIP_OK;
  const bool sync_failed = hip_check(hipStreamSynchronize(ln.st));
  c->dd_wgs_in_flight[ln.id] = 0;
  for (size_t b = 0; b < stt.who.size(); ++b) c->dd_open[stt.handles[stt.who[b]]].in_flight = false;
  stt.launched = false;
  if (sync_failed || !stt.complete) return DAFS_HIP_ELAUNCH;  // a launch that failed half-way: its nodes stay unfinished
  for (size_t b = 0; b < stt.who.size(); ++b) {
    const uint32_t h = stt.handles[stt.who[b]];
    const bool done = stt.paused[b] == 0;
    if (stt.paused[b] == 2 && !c->dd_open[h].no_split) { c->dd_open[h].no_split = true; ++c->dd_demotions; }  // its folders were lost: from now on the one-workgroup form
    c->dd_open[h].finished = done;
    if (done) c->dd_open[h].result.assign(stt.packed + stt.off[b], stt.packed + stt.off[b + 1]);
    if (finished) finished[stt.who[b]] = done ? 1 : 0;
  }
  return DAFS_HIP_OK;
}

int nodes_advance(dafs_hip_ctx* c, uint32_t n, const uint32_t* handles, dd_params dp, uint32_t max_iterations, uint8_t* finished) {
  advance_state stt;
  const dd_lane ln = lane_of(c, 0);
  const int rc = advance_launch(c, ln, n, handles, dp, max_iterations, finished, stt);
  const int rc2 = advance_collect(c, ln, stt, finished);  // also after a failure: whatever was enqueued is waited for
  return rc ? rc : rc2;
}

int nodes_result(dafs_hip_ctx* c, uint32_t handle, dafs_node_output* out, bool stamps) {
  if (handle >= c->dd_open.size() || !c->dd_open[handle].finished || c->dd_open[handle].released || c->dd_open[handle].in_flight) return DAFS_HIP_EINVAL;
  const dd_node& nd = c->dd_open[handle].nd;
  uint32_t info[16];
  float score = 0.0f;
  // x, y, z, score and info were carved back to back (nodes_open): one copy brings them all
  const uint8_t* lo = (const uint8_t*)nd.x;
  const uint8_t* hi = (const uint8_t*)(nd.info + 16);
  if (hi <= lo || (size_t)(hi - lo) > ((size_t)2 * nd.L1 + nd.L2 + 64) * 4 + 8 * 256) return DAFS_HIP_EINVAL;
  std::vector<uint8_t> blob((size_t)(hi - lo));
  const std::vector<uint32_t>& pre = c->dd_open[handle].result;
  if (pre.size() * 4 == blob.size()) memcpy(blob.data(), pre.data(), blob.size());  // came along with the launch the node finished in
  else if (hip_check(hipMemcpyAsync(blob.data(), lo, blob.size(), hipMemcpyDeviceToHost, c->stream)) || hip_check(hipStreamSynchronize(c->stream)))
    return DAFS_HIP_ELAUNCH;
  auto at = [&](const void* dev_ptr) { return blob.data() + ((const uint8_t*)dev_ptr - lo); };
  if (out->x) memcpy(out->x, at(nd.x), (size_t)nd.L1 * 4);
  if (out->y) memcpy(out->y, at(nd.y), (size_t)nd.L2 * 4);
  if (out->z) memcpy(out->z, at(nd.z), (size_t)nd.L1 * 4);
  memcpy(&score, at(nd.score), 4);
  memcpy(info, at(nd.info), sizeof info);
  if (stamps && nd.fold_fast & (16u | 32u | 64u | 128u)) {
    uint32_t sy[8] = {0};
    if (!hip_check(hipMemcpy(sy, nd.sync, sizeof sy, hipMemcpyDeviceToHost)))
      fprintf(stderr, "dd node L1=%u L2=%u folders | us: x-dp %.0f y-dp %.0f tracebacks %.0f\n", nd.L1, nd.L2, sy[5] / 100.0, sy[6] / 100.0, sy[7] / 100.0);
  }
  if (stamps)
    fprintf(stderr, "dd node L1=%u L2=%u n=%u+%u ncbp=%u iters=%u slow-xy=%u+%u | us: x-dp %.0f x-traceback %.0f wait %.0f cbp %.0f update %.0f tail %.0f | y %.0f z %.0f flags %x\n", nd.L1,
            nd.L2, nd.n1, nd.n2, info[0], info[1], info[4], info[5], info[8] / 100.0, info[9] / 100.0, info[10] / 100.0, info[11] / 100.0, info[12] / 100.0,
            info[13] / 100.0, info[14] / 100.0, info[15] / 100.0, nd.lds_flags);
  out->score = score;
  out->ncbp = info[0];
  out->iterations = info[1];
  out->violated = info[2];
  {  // The node's device memory is free for the nodes opened from now on.  Nothing can still use it: a node only becomes
     // `finished` in advance_collect, after its lane's stream has been drained, and a finished node is in no later launch.
     // No other open node may lie in the range (the free list would hand it out a second time).
    dafs_hip_ctx::dd_open_node& on = c->dd_open[handle];
    for (int k = 0; k < 2; ++k) {
      if (on.blk[k] && c->dd_range_live(on.blk[k], on.blk_bytes[k], &on)) {
        fprintf(stderr, "dafs_hip: node %u shares device memory with another open node\n", handle);
        return DAFS_HIP_ELAUNCH;
      }
      c->dd_free(on.blk[k], on.blk_bytes[k]);
    }
    on.released = true;
  }
  return info[3] ? DAFS_HIP_ELAUNCH : DAFS_HIP_OK;  // info[3]: the alignment traceback left the envelope
}

}  // namespace

// ---- resident nodes: the progressive phase without level barriers --------------------------------
// A node opened here stays on the device until dafs_hip_nodes_close.  dafs_hip_nodes_advance runs at
// most max_iterations further subgradient iterations of every listed node in ONE launch and reports
// which of them have finished; unfinished nodes simply take part in the next call, next to whatever
// nodes became ready in the meantime.  A node's results do not depend on how its iterations were cut
// into launches.
extern "C" int dafs_hip_nodes_open(dafs_hip_ctx* c, uint32_t nnodes, const dafs_node_input* in, const dafs_dd_params* prm, uint32_t* handles) {
  if (!c || !in || !prm || !handles || nnodes == 0) return DAFS_HIP_EINVAL;
  if (hip_check(hipSetDevice(c->device))) return DAFS_HIP_ENODEV;
  const uint32_t first = (uint32_t)c->dd_open.size();
  const int rc = nodes_open(c, lane_of(c, 0), nnodes, in, device_params(prm));
  if (rc) return rc;
  for (uint32_t b = 0; b < nnodes; ++b) handles[b] = first + b;
  return DAFS_HIP_OK;
}

extern "C" int dafs_hip_nodes_advance(dafs_hip_ctx* c, uint32_t n, const uint32_t* handles, const dafs_dd_params* prm, uint32_t max_iterations,
                                      uint8_t* finished) {
  if (!c || !handles || !prm || n == 0) return DAFS_HIP_EINVAL;
  if (hip_check(hipSetDevice(c->device))) return DAFS_HIP_ENODEV;
  return nodes_advance(c, n, handles, device_params(prm), max_iterations, finished);
}

// One round of the progressive phase in a single call: the open nodes advance (lane 0) while the nodes whose children have
// just finished are set up and started beside them (lane 1) -- their set-up kernels (averages, lists, constraints: ~0.8 ms
// a call, one or two workgroups busy) no longer stand between two launches of the solver.  With budget_us > 0 every node of
// the round also stops at the same moment (dd_params::budget), so the late starters do not stretch the round.
extern "C" int dafs_hip_nodes_round(dafs_hip_ctx* c, uint32_t n_new, const dafs_node_input* in, uint32_t* new_handles, uint32_t n_old,
                                    const uint32_t* old_handles, const dafs_dd_params* prm, uint32_t max_iterations, uint32_t budget_us,
                                    uint8_t* finished_old, uint8_t* finished_new) {
  if (!c || !prm || (n_new && (!in || !new_handles)) || (n_old && !old_handles) || (n_new == 0 && n_old == 0)) return DAFS_HIP_EINVAL;
  if (hip_check(hipSetDevice(c->device))) return DAFS_HIP_ENODEV;
  dd_params dp = device_params(prm);
  dp.budget = (unsigned long long)budget_us * 100ull;  // wall_clock64 ticks at 100 MHz
  int rc = DAFS_HIP_OK;
  advance_state st_old, st_new;
  const dd_lane l0 = lane_of(c, 0);
  // without open nodes there is nothing to overlap with: everything on the main lane
  const dd_lane l1 = n_old ? lane_of(c, 1) : l0;
  if (n_old) {
    if (dp.budget) {
      if ((rc = c->d_tref.reserve(1))) return rc;
      if (hip_check(hipMemsetAsync(c->d_tref.ptr, 0, sizeof(unsigned long long), l0.st))) return DAFS_HIP_ELAUNCH;
      dp.t_ref = c->d_tref.ptr;
      dp.t_ref_write = 1;
    }
    if ((rc = advance_launch(c, l0, n_old, old_handles, dp, max_iterations, finished_old, st_old))) {
      (void)advance_collect(c, l0, st_old, finished_old);  // whatever part of the launch was enqueued is waited for
      return rc;
    }
  }
  if (n_new) {
    const uint32_t first = (uint32_t)c->dd_open.size();
    dd_params dpn = dp;
    dpn.t_ref_write = 0;  // a late starter takes the round's reference tick (none when it runs alone)
    if (!n_old) dpn.t_ref = nullptr;
    rc = nodes_open(c, l1, n_new, in, dpn);
    if (rc) { (void)advance_collect(c, l0, st_old, finished_old); return rc; }
    for (uint32_t b = 0; b < n_new; ++b) new_handles[b] = first + b;
    rc = advance_launch(c, l1, n_new, new_handles, dpn, max_iterations, finished_new, st_new);
  }
  const int rc0 = advance_collect(c, l0, st_old, finished_old);
  const int rc1 = n_new ? advance_collect(c, l1, st_new, finished_new) : DAFS_HIP_OK;
  return rc ? rc : (rc0 ? rc0 : rc1);
}

extern "C" int dafs_hip_nodes_result(dafs_hip_ctx* c, uint32_t handle, dafs_node_output* out) {
  if (!c || !out) return DAFS_HIP_EINVAL;
  if (hip_check(hipSetDevice(c->device))) return DAFS_HIP_ENODEV;
  return nodes_result(c, handle, out, getenv("DAFS_HIP_DD_STAMPS") != nullptr);
}

extern "C" int dafs_hip_nodes_close(dafs_hip_ctx* c) {
  if (!c) return DAFS_HIP_EINVAL;
  if (hip_check(hipSetDevice(c->device))) return DAFS_HIP_ENODEV;
  if (hip_check(hipStreamSynchronize(c->stream))) return DAFS_HIP_ELAUNCH;
  c->dd_reset();
  return DAFS_HIP_OK;
}

extern "C" int dafs_hip_nodes_demotions(dafs_hip_ctx* c, uint32_t* n) {
  if (!c || !n) return DAFS_HIP_EINVAL;
  *n = c->dd_demotions;
  return DAFS_HIP_OK;
}

extern "C" int dafs_hip_nodes_memory(dafs_hip_ctx* c, uint64_t* reserved, uint64_t* in_use, uint64_t* peak) {
  if (!c) return DAFS_HIP_EINVAL;
  uint64_t r = 0;
  for (const dafs_hip_ctx::dd_chunk& ch : c->dd_chunks) r += ch.cap;
  if (reserved) *reserved = r;
  if (in_use) *in_use = c->dd_in_use;
  if (peak) *peak = c->dd_peak;
  return DAFS_HIP_OK;
}

// One batch of independent nodes, start to finish (the level-synchronous form; also what the refinement
// steps use).  Not to be mixed with open resident nodes.
extern "C" int dafs_hip_solve_nodes(dafs_hip_ctx* c, uint32_t nnodes, const dafs_node_input* in, const dafs_dd_params* prm,
                                    dafs_node_output* out) {
  if (!c || !in || !prm || !out || nnodes == 0) return DAFS_HIP_EINVAL;
  if (hip_check(hipSetDevice(c->device))) return DAFS_HIP_ENODEV;
  if (!c->dd_open.empty()) return DAFS_HIP_EINVAL;
  const dd_params dp = device_params(prm);
  int rc = nodes_open(c, lane_of(c, 0), nnodes, in, dp);
  std::vector<uint32_t> handles(nnodes);
  for (uint32_t b = 0; b < nnodes; ++b) handles[b] = b;
  // one launch runs every node to its end -- unless a split node lost its folding workgroups and was parked for the
  // one-workgroup form (k_dd_solve): then the unfinished nodes go round again
  std::vector<uint8_t> fin(nnodes, 0);
  for (int round = 0; !rc && round < 4; ++round) {
    rc = nodes_advance(c, nnodes, handles.data(), dp, 0, fin.data());
    if (std::all_of(fin.begin(), fin.end(), [](uint8_t f) { return f != 0; })) break;
  }
  if (!rc && !std::all_of(fin.begin(), fin.end(), [](uint8_t f) { return f != 0; })) rc = DAFS_HIP_ELAUNCH;
  for (uint32_t b = 0; b < nnodes && !rc; ++b) rc = nodes_result(c, b, &out[b], dp.stamps != 0);
  (void)hipStreamSynchronize(c->stream);
  c->dd_reset();
  return rc;
}

// Averaged base-pairing matrix of an alignment and its MEA structure: the final step of
// DAFS::run (dafs.cpp:1857-1871) without the RNAalifold term.  p_out (len*len,
// optional) receives the averaged matrix.
// bps: the store to average; by_row: its index is the alignment row (dafs_hip_update_basepairing) instead of the sequence.
// decode = false stops after the average (ss / score untouched).
static int average_and_decode(dafs_hip_ctx* c, uint32_t n, uint32_t len, const uint32_t* seq, const uint8_t* mask, const bp_store& bps, bool by_row,
                              bool decode, float th, uint32_t* ss, float* score, float* p_out) {
  if (!c || !n || !len || !seq || !mask || (decode && !ss)) return DAFS_HIP_EINVAL;
  if (hip_check(hipSetDevice(c->device))) return DAFS_HIP_ENODEV;
  if (!bps.valid) return DAFS_HIP_EINVAL;
  geom g;
  int rc;
  if ((rc = make_geom(c, n, len, seq, mask, g))) return rc;
  dd_node nd;
  carver cv;
  const size_t LL = (size_t)len * len;
  uint32_t* d_ss = nullptr;
  for (int pass = 0; pass < 2; ++pass) {
    cv.used = 0;
    memset(&nd, 0, sizeof nd);
    nd.n1 = n; nd.L1 = len;
    nd.seq1 = cv.take<uint32_t>(n);
    nd.rank1 = cv.take<uint32_t>((size_t)n * len);
    nd.idx1 = cv.take<uint32_t>(g.idx.size() + 1);
    nd.idxoff1 = cv.take<uint32_t>(n);
    nd.p_x = cv.take<float>(LL);
    carve_nuss(cv, len, nd.wx);
    d_ss = cv.take<uint32_t>((size_t)len + 1);
    nd.score = cv.take<float>(1);
    if (pass == 0) {
      if ((rc = c->work.reserve(cv.used + 256))) return rc;
      cv.base = c->work.ptr;
    }
  }
  if (hip_check(hipMemsetAsync(nd.p_x, 0, LL * 4, c->stream))) return DAFS_HIP_ELAUNCH;
  std::vector<uint32_t> rows(n);
  for (uint32_t r = 0; r < n; ++r) rows[r] = by_row ? r : seq[r];
  if (hip_check(hipMemcpyAsync((void*)nd.seq1, rows.data(), n * 4, hipMemcpyHostToDevice, c->stream))) return DAFS_HIP_ELAUNCH;
  if (hip_check(hipMemcpyAsync((void*)nd.rank1, g.rank.data(), g.rank.size() * 4, hipMemcpyHostToDevice, c->stream))) return DAFS_HIP_ELAUNCH;
  if (hip_check(hipMemcpyAsync((void*)nd.idx1, g.idx.data(), g.idx.size() * 4, hipMemcpyHostToDevice, c->stream))) return DAFS_HIP_ELAUNCH;
  if (hip_check(hipMemcpyAsync((void*)nd.idxoff1, g.idxoff.data(), n * 4, hipMemcpyHostToDevice, c->stream))) return DAFS_HIP_ELAUNCH;
  if ((rc = c->d_nodes.upload(&nd, 1, c->stream))) return rc;
  mp_store_dev none;
  memset(&none, 0, sizeof none);
  if ((rc = dd_avg_launch(c->d_nodes.ptr, 1, len, none, bps.view(), 0, 0, c->stream))) return rc;
  float s = 0;
  if (decode) {
    if ((rc = nussinov_launch(len, nd.p_x, nullptr, 0.0f, th, nd.wx, d_ss, nd.score, c->stream))) return rc;
    if (hip_check(hipMemcpyAsync(ss, d_ss, (size_t)len * 4, hipMemcpyDeviceToHost, c->stream))) return DAFS_HIP_ELAUNCH;
    if (hip_check(hipMemcpyAsync(&s, nd.score, 4, hipMemcpyDeviceToHost, c->stream))) return DAFS_HIP_ELAUNCH;
  }
  if (p_out && hip_check(hipMemcpyAsync(p_out, nd.p_x, LL * 4, hipMemcpyDeviceToHost, c->stream))) return DAFS_HIP_ELAUNCH;
  if (hip_check(hipStreamSynchronize(c->stream))) return DAFS_HIP_ELAUNCH;
  if (score) *score = s;
  return DAFS_HIP_OK;
}

extern "C" int dafs_hip_consensus_structure(dafs_hip_ctx* c, uint32_t n, uint32_t len, const uint32_t* seq, const uint8_t* mask,
                                            float th, uint32_t* ss, float* score, float* p_out) {
  if (!c) return DAFS_HIP_EINVAL;
  return average_and_decode(c, n, len, seq, mask, c->bp[c->cur_bp], false, true, th, ss, score, p_out);
}

// DAFS::update_basepairing_probability (dafs.cpp:609-712, options --bp-update / --bp-update1; no RNAalifold term, one level
// of brackets): every sequence of the alignment is folded again under the constraint that the common structure ss puts on
// it -- paired columns whose two residues exist in the row become '(' and ')', everything else stays free -- and the
// constrained posteriors (> CUTOFF) are averaged over the rows like the unconstrained ones, cut off at CUTOFF.
// The constraint strings are host work (index mapping); the folds run as one batch, the average on the device.
extern "C" int dafs_hip_update_basepairing(dafs_hip_ctx* c, uint32_t n, uint32_t len, const uint32_t* seq, const uint8_t* mask,
                                           const uint32_t* ss, float* p_out) {
  if (!c || !n || !len || !seq || !mask || !ss || !p_out) return DAFS_HIP_EINVAL;
  if (hip_check(hipSetDevice(c->device))) return DAFS_HIP_ENODEV;
  std::vector<char> str((size_t)len + 1);
  dafs_hip_make_brackets(len, ss, str.data());
  std::vector<std::string> cons(n);
  std::vector<uint32_t> rev(len);
  for (uint32_t r = 0; r < n; ++r) {
    if (seq[r] >= c->len.size()) return DAFS_HIP_EINVAL;
    uint32_t k = 0;
    for (uint32_t i = 0; i < len; ++i) rev[i] = mask[(size_t)r * len + i] ? k++ : DAFS_HIP_NONE;
    if (k != c->len[seq[r]]) return DAFS_HIP_EINVAL;
    std::string& con = cons[r];
    con.assign(k, '?');
    for (uint32_t i = 0; i < len; ++i)
      if (ss[i] != DAFS_HIP_NONE && ss[i] < len && rev[i] != DAFS_HIP_NONE && rev[ss[i]] != DAFS_HIP_NONE) {  // :640-652
        if (str[i] == '(') { con[rev[i]] = '('; con[rev[ss[i]]] = ')'; }
        else { con[rev[i]] = '.'; con[rev[ss[i]]] = '.'; }
      }
  }
  int rc = dafs_fold_rows_constrained(c, n, seq, cons, 0.01f, c->bp_rows);  // CONTRAfold(CUTOFF), dafs.cpp:1704
  if (rc) return rc;
  return average_and_decode(c, n, len, seq, mask, c->bp_rows, true, false, 0.0f, nullptr, nullptr, p_out);
}
