// dafs_amd/csrc/dd.hip -- the progressive phase on the GPU: posterior averaging, the consensus
// base-pair structure, and the dual-decomposition loop with its three DP subproblems.
//
// Reference functions replaced (paths relative to /root/reference):
//   average_basepairing_probability / average_matching_probability   src/dafs.cpp:513-607
//   DAFS::solve_by_dd (cbp enumeration, multiplier updates, step size) src/dafs.cpp:1006-1295
//   SparseNussinov::decode (both overloads)                           src/nussinov.cpp:207-392
//   SparseNeedlemanWunsch::initialize / decode (both overloads)       src/needleman_wunsch.cpp:198-422
//
// One workgroup solves one guide-tree node; independent nodes of a batch run side by side.  The
// whole subgradient loop (up to t_max iterations) stays on the device: per iteration the two
// Nussinov tables advance together one span-diagonal per barrier, the alignment table one
// anti-diagonal per barrier, the three tracebacks run on three different wavefronts, and the
// multiplier update is parallel over the sparse consensus structure.  Float sums that the
// reference forms sequentially (the dual value s, which steers the step size) are formed in the
// same order: positive terms are compacted in consensus-pair order and added by one lane.
#include <hip/hip_runtime.h>
#include <float.h>
#include <stdint.h>
#include "../../include/dafs_hip.h"
#include "dd.h"
#include "hip_util.h"

namespace dafs {

#define DD_CUTOFF 0.01f  // reference CUTOFF (double 0.01): for float v, v > 0.01 <=> v > 0.01f
#define DD_NONE 0xFFFFFFFFu
#define DD_THREADS 512

// ------------------------------------------------------------------------------------------
// SparseNussinov
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void nuss_cell(uint32_t L, uint32_t i, uint32_t j, const float* __restrict__ p,
                                          const float* __restrict__ q, float w, float th, const nuss_ws& ws) {
  float* dp = ws.dp;
  float v = 0.0f;
  uint32_t t = 0;
  if (i + 1 < j) { v = dp[(size_t)(i + 1) * L + j]; t = 1; }
  if (i < j - 1 && v < dp[(size_t)i * L + j - 1]) { v = dp[(size_t)i * L + j - 1]; t = 2; }
  uint32_t n = ws.cc[j];
  if (i + 1 < j - 1) {
    const float pij = p[(size_t)i * L + j];
    const float s = q ? w * (pij - th) - q[(size_t)i * L + j] : pij - th;  // nussinov.cpp:236 / :329
    if (s > 0.0f) {
      const float c = dp[(size_t)(i + 1) * L + j - 1] + s;
      ws.ck[(size_t)j * L + n] = i;
      ws.cv[(size_t)j * L + n] = c;
      ws.cc[j] = n + 1;  // one cell per column and diagonal: no other lane touches column j now
      if (v < c) { v = c; t = 3; }
    }
  }
  for (uint32_t x = 0; x < n; ++x) {  // earlier candidates of column j: all have k > i (:247-259)
    const uint32_t k = ws.ck[(size_t)j * L + x];
    const float c = dp[(size_t)i * L + k - 1] + ws.cv[(size_t)j * L + x];
    if (v < c) { v = c; t = k - i + 3; }
  }
  dp[(size_t)i * L + j] = v;
  ws.tr[(size_t)i * L + j] = t;
}

// traceback with an explicit stack (:265-295); `stack` has room for 2*(L+2) pairs
__device__ void nuss_traceback(uint32_t L, const nuss_ws& ws, uint32_t* ss, uint32_t* stack) {
  uint32_t sp = 0;
  stack[0] = 0; stack[1] = L - 1; sp = 1;
  uint32_t guard = 4 * L + 8;
  while (sp && guard--) {
    --sp;
    const int i = (int)stack[2 * sp], j = (int)stack[2 * sp + 1];
    const uint32_t t = ws.tr[(size_t)i * L + j];
    if (t == 0) continue;
    if (t == 1) { stack[2 * sp] = i + 1; stack[2 * sp + 1] = j; ++sp; }
    else if (t == 2) { stack[2 * sp] = i; stack[2 * sp + 1] = j - 1; ++sp; }
    else if (t == 3) { ss[i] = j; stack[2 * sp] = i + 1; stack[2 * sp + 1] = j - 1; ++sp; }
    else {
      const int k = i + (int)t - 3;
      stack[2 * sp] = i; stack[2 * sp + 1] = k - 1; ++sp;
      ss[k] = j;
      stack[2 * sp] = k + 1; stack[2 * sp + 1] = j - 1; ++sp;
    }
  }
}

// Two independent problems (A and B; LB may be 0) advance one span per barrier.
__device__ void nuss_pair_dp(uint32_t LA, const float* pA, const float* qA, float wA, const nuss_ws& A,
                             uint32_t LB, const float* pB, const float* qB, float wB, const nuss_ws& B, float th) {
  const uint32_t tid = threadIdx.x, nt = blockDim.x;
  for (uint32_t i = tid; i < LA; i += nt) { A.dp[(size_t)i * LA + i] = 0.0f; A.tr[(size_t)i * LA + i] = 0; A.cc[i] = 0; }
  for (uint32_t i = tid; i < LB; i += nt) { B.dp[(size_t)i * LB + i] = 0.0f; B.tr[(size_t)i * LB + i] = 0; B.cc[i] = 0; }
  __syncthreads();
  const uint32_t Lm = LA > LB ? LA : LB;
  for (uint32_t l = 1; l < Lm; ++l) {
    const uint32_t na = l < LA ? LA - l : 0, nb = l < LB ? LB - l : 0;
    for (uint32_t c = tid; c < na + nb; c += nt) {
      if (c < na) nuss_cell(LA, c, c + l, pA, qA, wA, th, A);
      else nuss_cell(LB, c - na, c - na + l, pB, qB, wB, th, B);
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------
// SparseNeedlemanWunsch
// ------------------------------------------------------------------------------------------
// initialize (needleman_wunsch.cpp:198-253); fa/la = scratch of L1+1 uint32 each
__device__ void nw_envelope(uint32_t L1, uint32_t L2, const float* __restrict__ p, float th, uint32_t* env,
                            uint32_t* fa, uint32_t* la) {
  const uint32_t tid = threadIdx.x, nt = blockDim.x;
  for (uint32_t i = tid + 1; i <= L1; i += nt) {
    uint32_t f = 0, l = 0;
    for (uint32_t k = 1; k <= L2; ++k)
      if (p[(size_t)(i - 1) * L2 + (k - 1)] - th >= 0.0f) { f = k; break; }
    if (f)
      for (uint32_t k = L2; k != 0; --k)
        if (p[(size_t)(i - 1) * L2 + (k - 1)] - th >= 0.0f) { l = k; break; }
    fa[i] = f;
    la[i] = l;
  }
  __syncthreads();
  if (tid == 0) {
    for (uint32_t i = 0; i <= L1; ++i) { env[2 * i] = 0; env[2 * i + 1] = 0; }
    for (uint32_t i = 1; i <= L1; ++i) {
      if (fa[i]) {
        if (fa[i] - 1 < env[2 * (i - 1)]) env[2 * (i - 1)] = fa[i] - 1;
        env[2 * i] = fa[i];
      }
      if (env[2 * i] == 0) {
        env[2 * i] = env[2 * (i - 1)];
        env[2 * i + 1] = env[2 * (i - 1) + 1];
        continue;
      }
      if (la[i] - 1 > env[2 * (i - 1) + 1]) env[2 * (i - 1) + 1] = la[i] - 1;
      env[2 * i + 1] = la[i];
    }
    env[2 * L1 + 1] = L2;
    for (uint32_t i = L1, v = L2; i != 0; --i) { v = v < env[2 * i] ? v : env[2 * i]; env[2 * i] = v; }
    for (uint32_t i = 0, v = 0; i != L1 + 1; ++i) { v = v > env[2 * i + 1] ? v : env[2 * i + 1]; env[2 * i + 1] = v; }
    for (uint32_t i = 1; i != L1 + 1; ++i)
      if (env[2 * (i - 1) + 1] < env[2 * i]) env[2 * i] = env[2 * (i - 1) + 1];
  }
  __syncthreads();
}

// table initialisation (:262-274); cells inside the envelope are overwritten by every decode,
// cells outside keep lowest()/' ' forever, so this runs once per problem
__device__ void nw_init(uint32_t L1, uint32_t L2, float* dp, uint8_t* tr) {
  const uint32_t W = L2 + 1;
  for (size_t c = threadIdx.x; c < (size_t)(L1 + 1) * W; c += blockDim.x) {
    const uint32_t i = (uint32_t)(c / W), k = (uint32_t)(c % W);
    const bool border = (i == 0 || k == 0);
    dp[c] = border ? 0.0f : -FLT_MAX;
    tr[c] = (i == 0 && k == 0) ? ' ' : (k == 0 ? 'X' : (i == 0 ? 'Y' : ' '));
  }
  __syncthreads();
}

// decode DP (:276-296), one anti-diagonal per barrier
__device__ void nw_dp(uint32_t L1, uint32_t L2, const float* __restrict__ p, const float* __restrict__ q, float th,
                      const uint32_t* __restrict__ env, float* dp, uint8_t* tr) {
  const uint32_t W = L2 + 1;
  for (uint32_t d = 2; d <= L1 + L2; ++d) {
    const uint32_t ilo = d > L2 ? d - L2 : 1, ihi = d - 1 < L1 ? d - 1 : L1;
    for (uint32_t i = ilo + threadIdx.x; i <= ihi; i += blockDim.x) {
      const uint32_t k = d - i;
      if (k < env[2 * i] || k > env[2 * i + 1]) continue;
      float v = dp[(size_t)(i - 1) * W + (k - 1)] + p[(size_t)(i - 1) * L2 + (k - 1)] - th;
      if (q) v = v + q[(size_t)(i - 1) * L2 + (k - 1)];
      uint8_t t = 'M';
      if (v < dp[(size_t)(i - 1) * W + k]) { v = dp[(size_t)(i - 1) * W + k]; t = 'X'; }
      if (v < dp[(size_t)i * W + (k - 1)]) { v = dp[(size_t)i * W + (k - 1)]; t = 'Y'; }
      dp[(size_t)i * W + k] = v;
      tr[(size_t)i * W + k] = t;
    }
    __syncthreads();
  }
}

// traceback + decode of the path (:298-335), single lane; returns false if the path left the
// envelope (the reference would not terminate there)
__device__ bool nw_traceback(uint32_t L1, uint32_t L2, const uint8_t* tr, uint32_t* al) {
  const uint32_t W = L2 + 1;
  int i = (int)L1, k = (int)L2;
  uint32_t guard = L1 + L2 + 2;
  while ((i > 0 || k > 0) && guard--) {
    const uint8_t t = tr[(size_t)i * W + k];
    if (t == 'M') { al[i - 1] = (uint32_t)(k - 1); --i; --k; }
    else if (t == 'X') { al[i - 1] = DD_NONE; --i; }
    else if (t == 'Y') { --k; }
    else return false;
  }
  return i == 0 && k == 0;
}

// ------------------------------------------------------------------------------------------
// standalone decoders (Fold::Decoder / Align::Decoder plugin calls)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(DD_THREADS) void k_nussinov_single(uint32_t L, const float* p, const float* q, float w, float th,
                                                                nuss_ws ws, uint32_t* ss, float* score) {
  for (uint32_t i = threadIdx.x; i < L; i += blockDim.x) ss[i] = DD_NONE;
  nuss_ws none = {nullptr, nullptr, nullptr, nullptr, nullptr};
  nuss_pair_dp(L, p, q, w, ws, 0, nullptr, nullptr, 0.0f, none, th);
  if (threadIdx.x == 0) {
    nuss_traceback(L, ws, ss, ws.ck);  // the candidate-key array is free again: reuse it as the stack
    *score = ws.dp[L - 1];
  }
}

__global__ __launch_bounds__(DD_THREADS) void k_nw_single(uint32_t L1, uint32_t L2, const float* p, const float* q, float th,
                                                          uint32_t* env, int compute_env, float* dp, uint8_t* tr, uint32_t* al,
                                                          float* score) {
  if (compute_env) nw_envelope(L1, L2, p, th, env, al, (uint32_t*)dp);  // al / dp double as scratch before use
  nw_init(L1, L2, dp, tr);
  nw_dp(L1, L2, p, q, th, env, dp, tr);
  if (threadIdx.x == 0) {
    const bool ok = nw_traceback(L1, L2, tr, al);
    *score = ok ? dp[(size_t)L1 * (L2 + 1) + L2] : __builtin_nanf("");
  }
}

// ------------------------------------------------------------------------------------------
// posterior averaging: one workgroup per (node, matrix); every thread owns whole rows so each
// cell receives its addends in the reference's order (rows of aln1, then rows of aln2)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_node_avg(const dd_node* nodes, mp_store_dev mp, bp_store_dev bp) {
  const dd_node nd = nodes[blockIdx.x / 3];
  const uint32_t role = blockIdx.x % 3;
  if (role < 2) {  // average_basepairing_probability, dafs.cpp:561-607 (no alifold term)
    const uint32_t L = role ? nd.L2 : nd.L1, n = role ? nd.n2 : nd.n1;
    const uint32_t* seq = role ? nd.seq2 : nd.seq1;
    const uint32_t* rank = role ? nd.rank2 : nd.rank1;
    const uint32_t* idx = role ? nd.idx2 : nd.idx1;
    const uint32_t* idxoff = role ? nd.idxoff2 : nd.idxoff1;
    float* P = role ? nd.p_y : nd.p_x;
    for (uint32_t I = threadIdx.x; I < L; I += blockDim.x) {
      float* row = P + (size_t)I * L;
      for (uint32_t r = 0; r < n; ++r) {
        const uint32_t ii = rank[(size_t)r * L + I];
        if (ii == DD_NONE) continue;
        const row_ref b = bp_row(bp, seq[r], ii);
        const uint32_t* map = idx + idxoff[r];
        for (uint32_t e = 0; e < b.n; ++e) row[map[b.col[e]]] += b.val[e] / n;
      }
      for (uint32_t J = I + 1; J < L; ++J)
        if (row[J] <= DD_CUTOFF) row[J] = 0.0f;
    }
  } else {  // average_matching_probability, dafs.cpp:513-559
    const uint32_t L1 = nd.L1, L2 = nd.L2;
    const uint32_t nn = nd.n1 * nd.n2;
    for (uint32_t I = threadIdx.x; I < L1; I += blockDim.x) {
      float* row = nd.p_z + (size_t)I * L2;
      for (uint32_t r1 = 0; r1 < nd.n1; ++r1) {
        const uint32_t ii = nd.rank1[(size_t)r1 * L1 + I];
        if (ii == DD_NONE) continue;
        const uint32_t s1 = nd.seq1[r1];
        for (uint32_t r2 = 0; r2 < nd.n2; ++r2) {
          const row_ref m = mp_row(mp, s1, nd.seq2[r2], ii);
          const uint32_t* map = nd.idx2 + nd.idxoff2[r2];
          for (uint32_t e = 0; e < m.n; ++e) row[map[m.col[e]]] += m.val[e] / nn;
        }
      }
      for (uint32_t J = 0; J < L2; ++J) {
        if (row[J] <= DD_CUTOFF) row[J] = 0.0f;
        if (row[J] > 1.0f) row[J] = 1.0f;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// sparse structure of one node: entry lists of p_x, p_y, p_z, consensus base-pair counts,
// alignment envelope, table initialisation.  xmap/ymap/zmap arrive filled with -1, q_* with 0.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ bool cbp_ok(const dd_node& nd, const dd_params& prm, float px, float py, float pzik, float pzjl) {
  const float p = (nd.n1 * px + nd.n2 * py) / (nd.n1 + nd.n2);  // dafs.cpp:1032
  const float q = (pzik + pzjl) / 2;                               // :1033
  return (p - prm.th_s > 0.0f) && (prm.w * (p - prm.th_s) + (q - prm.th_a) > 0.0f);
}

__device__ void row_lists(uint32_t R, uint32_t Cn, const float* P, bool upper, uint32_t* ptr, uint32_t* lst, int32_t* map) {
  const uint32_t tid = threadIdx.x, nt = blockDim.x;
  for (uint32_t i = tid; i < R; i += nt) {
    uint32_t c = 0;
    for (uint32_t j = upper ? i + 1 : 0; j < Cn; ++j) c += P[(size_t)i * Cn + j] > DD_CUTOFF ? 1 : 0;
    ptr[i + 1] = c;
  }
  __syncthreads();
  if (tid == 0) {
    ptr[0] = 0;
    for (uint32_t i = 0; i < R; ++i) ptr[i + 1] += ptr[i];
  }
  __syncthreads();
  for (uint32_t i = tid; i < R; i += nt) {
    uint32_t pos = ptr[i];
    for (uint32_t j = upper ? i + 1 : 0; j < Cn; ++j)
      if (P[(size_t)i * Cn + j] > DD_CUTOFF) {
        lst[pos] = j;
        if (map) map[(size_t)i * Cn + j] = (int32_t)pos;
        ++pos;
      }
  }
  __syncthreads();
}

__global__ __launch_bounds__(DD_THREADS) void k_node_lists(const dd_node* nodes, dd_params prm) {
  const dd_node nd = nodes[blockIdx.x];
  const uint32_t tid = threadIdx.x, nt = blockDim.x;
  const uint32_t L1 = nd.L1, L2 = nd.L2;
  row_lists(L1, L1, nd.p_x, true, nd.px_ptr, nd.px_j, nd.xmap);
  row_lists(L2, L2, nd.p_y, true, nd.py_ptr, nd.py_l, nd.ymap);
  row_lists(L1, L2, nd.p_z, false, nd.pz_ptr, nd.pz_k, nullptr);
  // consensus base pairs per p_x entry (dafs.cpp:1022-1044)
  __shared__ uint32_t s_total;
  if (tid == 0) s_total = 0;
  __syncthreads();
  uint32_t mine = 0;
  for (uint32_t i = 0; i < L1; ++i)
    for (uint32_t e = nd.px_ptr[i] + tid; e < nd.px_ptr[i + 1]; e += nt) {
      const uint32_t j = nd.px_j[e];
      const float px = nd.p_x[(size_t)i * L1 + j];
      uint32_t c = 0;
      for (uint32_t a = nd.pz_ptr[i]; a < nd.pz_ptr[i + 1]; ++a) {
        const uint32_t k = nd.pz_k[a];
        const float pzik = nd.p_z[(size_t)i * L2 + k];
        for (uint32_t b = nd.py_ptr[k]; b < nd.py_ptr[k + 1]; ++b) {
          const uint32_t l = nd.py_l[b];
          const float pzjl = nd.p_z[(size_t)j * L2 + l];
          if (pzjl > DD_CUTOFF && cbp_ok(nd, prm, px, nd.p_y[(size_t)k * L2 + l], pzik, pzjl)) ++c;
        }
      }
      nd.cbp_cnt[e] = c;
      mine += c;
    }
  atomicAdd(&s_total, mine);
  // alignment envelope + table initialisation
  nw_envelope(L1, L2, nd.p_z, prm.th_a, nd.env, nd.x, nd.z);  // x / z double as scratch here
  nw_init(L1, L2, nd.dp_z, nd.tr_z);
  __syncthreads();
  if (tid == 0) { nd.info[0] = s_total; nd.info[1] = 0; nd.info[2] = 0; nd.info[3] = 0; }
}

__global__ __launch_bounds__(DD_THREADS) void k_node_cbp_fill(const dd_node* nodes, dd_params prm) {
  const dd_node nd = nodes[blockIdx.x];
  const uint32_t tid = threadIdx.x, nt = blockDim.x;
  const uint32_t L1 = nd.L1, L2 = nd.L2;
  const uint32_t npx = nd.px_ptr[L1];
  // exclusive prefix of the per-entry counts (entries are already in (i,j) order)
  if (tid == 0) {
    uint32_t run = 0;
    for (uint32_t e = 0; e < npx; ++e) { const uint32_t c = nd.cbp_cnt[e]; nd.cbp_cnt[e] = run; run += c; }
  }
  __syncthreads();
  for (uint32_t i = 0; i < L1; ++i)
    for (uint32_t e = nd.px_ptr[i] + tid; e < nd.px_ptr[i + 1]; e += nt) {
      const uint32_t j = nd.px_j[e];
      const float px = nd.p_x[(size_t)i * L1 + j];
      uint32_t u = nd.cbp_cnt[e];
      const uint32_t u0 = u;
      for (uint32_t a = nd.pz_ptr[i]; a < nd.pz_ptr[i + 1]; ++a) {
        const uint32_t k = nd.pz_k[a];
        const float pzik = nd.p_z[(size_t)i * L2 + k];
        for (uint32_t b = nd.py_ptr[k]; b < nd.py_ptr[k + 1]; ++b) {
          const uint32_t l = nd.py_l[b];
          const float pzjl = nd.p_z[(size_t)j * L2 + l];
          if (pzjl > DD_CUTOFF && cbp_ok(nd, prm, px, nd.p_y[(size_t)k * L2 + l], pzik, pzjl)) {
            if (u < nd.ncbp_cap) {
              uint32_t* c = nd.cbp + (size_t)8 * u;
              c[0] = i; c[1] = j; c[2] = k; c[3] = l; c[4] = e; c[5] = b;
            }
            nd.cy_flag[b] = 1;                       // c_y (dafs.cpp:1039)
            nd.cz_flag[(size_t)i * L2 + k] = 1;      // c_z (:1040-1041)
            nd.cz_flag[(size_t)j * L2 + l] = 1;
            ++u;
          }
        }
      }
      nd.cx_flag[e] = u > u0 ? 1 : 0;                // c_x (:1038)
    }
  __syncthreads();
  // c_z as sorted row lists (:1056-1060) + dense id map
  for (uint32_t i = tid; i < L1; i += nt) {
    uint32_t c = 0;
    for (uint32_t k = 0; k < L2; ++k) c += nd.cz_flag[(size_t)i * L2 + k];
    nd.cz_ptr[i + 1] = c;
  }
  __syncthreads();
  if (tid == 0) {
    nd.cz_ptr[0] = 0;
    for (uint32_t i = 0; i < L1; ++i) nd.cz_ptr[i + 1] += nd.cz_ptr[i];
  }
  __syncthreads();
  for (uint32_t i = tid; i < L1; i += nt) {
    uint32_t pos = nd.cz_ptr[i];
    for (uint32_t k = 0; k < L2; ++k)
      if (nd.cz_flag[(size_t)i * L2 + k]) { nd.cz_k[pos] = k; nd.zmap[(size_t)i * L2 + k] = (int32_t)pos; ++pos; }
  }
  __syncthreads();
  const uint32_t ncbp = nd.info[0] < nd.ncbp_cap ? nd.info[0] : nd.ncbp_cap;
  for (uint32_t u = tid; u < ncbp; u += nt) {
    uint32_t* c = nd.cbp + (size_t)8 * u;
    c[6] = (uint32_t)nd.zmap[(size_t)c[0] * L2 + c[2]];
    c[7] = (uint32_t)nd.zmap[(size_t)c[1] * L2 + c[3]];
  }
}

// ------------------------------------------------------------------------------------------
// the subgradient loop, dafs.cpp:1066-1294
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(DD_THREADS) void k_dd_solve(const dd_node* nodes, dd_params prm) {
  const dd_node nd = nodes[blockIdx.x];
  const uint32_t tid = threadIdx.x, nt = blockDim.x;
  const uint32_t L1 = nd.L1, L2 = nd.L2;
  const uint32_t ncbp = nd.info[0];
  const uint32_t npx = nd.px_ptr[L1], npy = nd.py_ptr[L2], ncz = nd.cz_ptr[L1];
  const float w_x = prm.w * 2 * nd.n1 / (nd.n1 + nd.n2);  // dafs.cpp:1091
  const float w_y = prm.w * 2 * nd.n2 / (nd.n1 + nd.n2);  // :1092
  __shared__ uint32_t s_cnt[DD_THREADS];
  __shared__ uint32_t s_violated, s_npos;
  __shared__ int s_stop, s_bad;
  __shared__ float s_eta;
  float c = 0.0f, eta = prm.eta0, s_prev = 0.0f;  // meaningful in thread 0
  uint32_t t = 0, violated = 0;
  if (tid == 0) { s_eta = eta; s_bad = 0; }
  __syncthreads();

  for (t = 0; t != prm.t_max; ++t) {
    for (uint32_t i = tid; i < L1; i += nt) nd.x[i] = DD_NONE;
    for (uint32_t k = tid; k < L2; k += nt) nd.y[k] = DD_NONE;
    nuss_pair_dp(L1, nd.p_x, nd.q_x, w_x, nd.wx, L2, nd.p_y, nd.q_y, w_y, nd.wy, prm.th_s);
    nw_dp(L1, L2, nd.p_z, nd.q_z, prm.th_a, nd.env, nd.dp_z, nd.tr_z);
    // three tracebacks on three wavefronts; counters cleared meanwhile
    if (tid == 0) nuss_traceback(L1, nd.wx, nd.x, nd.wx.ck);
    if (tid == 64) nuss_traceback(L2, nd.wy, nd.y, nd.wy.ck);
    if (tid == 128 && !nw_traceback(L1, L2, nd.tr_z, nd.z)) s_bad = 1;
    if (tid >= 192) {
      for (uint32_t e = tid - 192; e < npx; e += nt - 192) nd.tx[e] = 0;
      for (uint32_t e = tid - 192; e < npy; e += nt - 192) nd.ty[e] = 0;
      for (uint32_t e = tid - 192; e < ncz; e += nt - 192) nd.tz[e] = 0;
    }
    if (tid == 0) s_violated = 0;
    __syncthreads();

    // consensus constraints (:1103-1117): counts by atomics, positive s_w compacted in order
    const uint32_t chunk = (ncbp + nt - 1) / nt;
    const uint32_t u0 = tid * chunk < ncbp ? tid * chunk : ncbp;
    const uint32_t u1 = u0 + chunk < ncbp ? u0 + chunk : ncbp;
    uint32_t npos = 0;
    for (uint32_t u = u0; u < u1; ++u) {
      const uint32_t* cb = nd.cbp + (size_t)8 * u;
      const float s_w = nd.q_x[(size_t)cb[0] * L1 + cb[1]] + nd.q_y[(size_t)cb[2] * L2 + cb[3]] -
                        nd.q_z[(size_t)cb[0] * L2 + cb[2]] - nd.q_z[(size_t)cb[1] * L2 + cb[3]];
      if (s_w > 0.0f) {
        ++npos;
        atomicAdd(&nd.tx[cb[4]], 1);
        atomicAdd(&nd.ty[cb[5]], 1);
        atomicAdd(&nd.tz[cb[6]], 1);
        atomicAdd(&nd.tz[cb[7]], 1);
      }
    }
    s_cnt[tid] = npos;
    __syncthreads();
    if (tid == 0) {
      uint32_t run = 0;
      for (uint32_t k = 0; k < nt; ++k) { const uint32_t v = s_cnt[k]; s_cnt[k] = run; run += v; }
      s_npos = run;
    }
    __syncthreads();
    {
      uint32_t pos = s_cnt[tid];
      for (uint32_t u = u0; u < u1; ++u) {
        const uint32_t* cb = nd.cbp + (size_t)8 * u;
        const float s_w = nd.q_x[(size_t)cb[0] * L1 + cb[1]] + nd.q_y[(size_t)cb[2] * L2 + cb[3]] -
                          nd.q_z[(size_t)cb[0] * L2 + cb[2]] - nd.q_z[(size_t)cb[1] * L2 + cb[3]];
        if (s_w > 0.0f) nd.sw[pos++] = s_w;
      }
    }
    __syncthreads();
    eta = s_eta;

    // multiplier updates (:1121-1254), every cell touched by exactly one lane
    uint32_t viol = 0;
    for (uint32_t i = tid; i < L1; i += nt) {
      const uint32_t j = nd.x[i];
      if (j != DD_NONE) {
        const int32_t id = nd.xmap[(size_t)i * L1 + j];
        const int tc = id >= 0 ? nd.tx[id] : 0;
        if (tc != 1) { ++viol; nd.q_x[(size_t)i * L1 + j] -= eta * (tc - 1); }
      }
      for (uint32_t e = nd.px_ptr[i]; e < nd.px_ptr[i + 1]; ++e) {
        if (!nd.cx_flag[e]) continue;
        const uint32_t jj = nd.px_j[e];
        const int tc = nd.tx[e];
        if (j != jj && tc != 0) { ++viol; nd.q_x[(size_t)i * L1 + jj] -= eta * tc; }
      }
      const uint32_t kz = nd.z[i];
      if (kz != DD_NONE) {
        const int32_t id = nd.zmap[(size_t)i * L2 + kz];
        const int tc = id >= 0 ? nd.tz[id] : 0;
        if (tc > 1) ++viol;
        const float v = nd.q_z[(size_t)i * L2 + kz] - eta * (1 - tc);
        nd.q_z[(size_t)i * L2 + kz] = (0.0f < v) ? v : 0.0f;
      }
      for (uint32_t e = nd.cz_ptr[i]; e < nd.cz_ptr[i + 1]; ++e) {
        const uint32_t kk = nd.cz_k[e];
        if (kz != kk) {
          const int tc = nd.tz[e];
          if (tc > 0) ++viol;
          const float v = nd.q_z[(size_t)i * L2 + kk] + eta * tc;
          nd.q_z[(size_t)i * L2 + kk] = (0.0f < v) ? v : 0.0f;
        }
      }
    }
    for (uint32_t k = tid; k < L2; k += nt) {
      const uint32_t l = nd.y[k];
      if (l != DD_NONE) {
        const int32_t id = nd.ymap[(size_t)k * L2 + l];
        const int tc = id >= 0 ? nd.ty[id] : 0;
        if (tc != 1) { ++viol; nd.q_y[(size_t)k * L2 + l] -= eta * (tc - 1); }
      }
      for (uint32_t e = nd.py_ptr[k]; e < nd.py_ptr[k + 1]; ++e) {
        if (!nd.cy_flag[e]) continue;
        const uint32_t ll = nd.py_l[e];
        const int tc = nd.ty[e];
        if (l != ll && tc != 0) { ++viol; nd.q_y[(size_t)k * L2 + ll] -= eta * tc; }
      }
    }
    if (viol) atomicAdd(&s_violated, viol);
    __syncthreads();

    if (tid == 0) {
      // dual value in the reference's summation order (:1090-1093, :1111)
      float s = 0.0f;
      s += nd.wx.dp[L1 - 1];
      s += nd.wy.dp[L2 - 1];
      s += nd.dp_z[(size_t)L1 * (L2 + 1) + L2];
      const uint32_t np = s_npos;
      for (uint32_t k = 0; k < np; ++k) s += nd.sw[k];
      violated = s_violated;
      int stop = 0;
      if (violated == 0 && !prm.force_iters) stop = 1;  // :1278
      else {
        if (s > s_prev || t == 0) {                     // :1283-1288
          float num = 4.0f * ncbp - violated;
          num = (0.0f < num) ? num : 0.0f;
          c += num / (4.0 * ncbp);
          eta = prm.eta0 / (1.0 + c);
          s_eta = eta;
        }
        s_prev = s;
      }
      if (s_bad) stop = 1;
      s_stop = stop;
    }
    __syncthreads();
    if (s_stop) break;
  }
  if (tid == 0) {
    *nd.score = s_prev;
    nd.info[1] = t;
    nd.info[2] = violated;
    nd.info[3] = s_bad ? 1u : 0u;
  }
}

// ------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------
int dd_avg_launch(const dd_node* d_nodes, uint32_t nnodes, mp_store_dev mp, bp_store_dev bp, hipStream_t st) {
  if (!nnodes) return DAFS_HIP_OK;
  hipLaunchKernelGGL(k_node_avg, dim3(nnodes * 3), dim3(256), 0, st, d_nodes, mp, bp);
  return hip_check(hipGetLastError()) ? DAFS_HIP_ELAUNCH : DAFS_HIP_OK;
}
int dd_lists_launch(const dd_node* d_nodes, uint32_t nnodes, dd_params prm, hipStream_t st) {
  if (!nnodes) return DAFS_HIP_OK;
  hipLaunchKernelGGL(k_node_lists, dim3(nnodes), dim3(DD_THREADS), 0, st, d_nodes, prm);
  return hip_check(hipGetLastError()) ? DAFS_HIP_ELAUNCH : DAFS_HIP_OK;
}
int dd_cbp_fill_launch(const dd_node* d_nodes, uint32_t nnodes, dd_params prm, hipStream_t st) {
  if (!nnodes) return DAFS_HIP_OK;
  hipLaunchKernelGGL(k_node_cbp_fill, dim3(nnodes), dim3(DD_THREADS), 0, st, d_nodes, prm);
  return hip_check(hipGetLastError()) ? DAFS_HIP_ELAUNCH : DAFS_HIP_OK;
}
int dd_solve_launch(const dd_node* d_nodes, uint32_t nnodes, dd_params prm, hipStream_t st) {
  if (!nnodes) return DAFS_HIP_OK;
  hipLaunchKernelGGL(k_dd_solve, dim3(nnodes), dim3(DD_THREADS), 0, st, d_nodes, prm);
  return hip_check(hipGetLastError()) ? DAFS_HIP_ELAUNCH : DAFS_HIP_OK;
}
int nussinov_launch(uint32_t L, const float* p, const float* q, float w, float th, nuss_ws ws, uint32_t* ss, float* score, hipStream_t st) {
  hipLaunchKernelGGL(k_nussinov_single, dim3(1), dim3(DD_THREADS), 0, st, L, p, q, w, th, ws, ss, score);
  return hip_check(hipGetLastError()) ? DAFS_HIP_ELAUNCH : DAFS_HIP_OK;
}
int nw_launch(uint32_t L1, uint32_t L2, const float* p, const float* q, float th, uint32_t* env, int compute_env,
              float* dp, uint8_t* tr, uint32_t* al, float* score, hipStream_t st) {
  hipLaunchKernelGGL(k_nw_single, dim3(1), dim3(DD_THREADS), 0, st, L1, L2, p, q, th, env, compute_env, dp, tr, al, score);
  return hip_check(hipGetLastError()) ? DAFS_HIP_ELAUNCH : DAFS_HIP_OK;
}

}  // namespace dafs
