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
// Single-wavefront forms of the two DPs, used inside the subgradient loop.  Lane t owns W
// consecutive columns and keeps the previous row of its columns in LDS (P[c*64+lane]); rows are
// skewed by lane, the boundary column travels to the next lane by shuffle.  No barriers: the three
// subproblems of an iteration run concurrently on three wavefronts of the workgroup.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float ld_l2(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_l2(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ size_t tri_index(uint32_t L, uint32_t i, uint32_t j) { return (size_t)i * L - (size_t)i * (i - 1) / 2 + (j - i); }  // j >= i

// SparseNussinov::decode DP (nussinov.cpp:217-263) by one wavefront.  Rows run from L-1 down to 0
// (row i at step (L-1-i)+lane), which visits the cells of a column in the same order as the
// reference's span loop, so the per-column candidate lists are built in the same order.
// trb: one byte per upper-triangle cell (0..3, 4 = bifurcation with k in trk); dp/ck/cv in HBM
// (dp read back for the bifurcation terms through L2); cc, P in LDS.
#define DD_WMAX 16  // columns per lane in the wave DPs: sequences up to 64*16 = 1024 columns

// S: pair scores w*(p-th)-q (or p-th), precomputed by the whole workgroup (dd_fill_scores).  Row i of S
// is fetched one step ahead into registers (16 independent loads) and parked in LDS (Sb) for the next step,
// so no global-memory latency sits on the per-cell dependency chain.
__device__ float nuss_wave(uint32_t L, const float* __restrict__ S, const nuss_ws& ws, uint8_t* trb, uint32_t* trk, float* P, float* Sb,
                           uint32_t* cc, int lane) {
  const uint32_t W = (L + 63) / 64;
  for (uint32_t c = 0; c < W; ++c) {
    P[c * 64 + lane] = 0.0f;
    Sb[c * 64 + lane] = 0.0f;
    const uint32_t j = lane * W + c;
    if (j < L) cc[j] = 0;
  }
  float last = 0.0f, leftprev = 0.0f, score = 0.0f;
  const int nsteps = (int)L + 63;
  for (uint32_t c = 0; c < W; ++c) Sb[c * 64 + lane] = S[(size_t)c * 64 + lane];  // step 0
  for (int s = 0; s < nsteps; ++s) {
    const int i = (int)L - 1 - (s - lane);
    const bool rowv = i >= 0 && i < (int)L;
    // prefetch the scores of the next step (coalesced: S is stored in sweep order)
    float nxt[DD_WMAX];
    const bool nv = s + 1 < nsteps;
#pragma unroll
    for (int c = 0; c < DD_WMAX; ++c) nxt[c] = (nv && (uint32_t)c < W) ? S[((size_t)(s + 1) * W + c) * 64 + lane] : 0.0f;
    float recv = __shfl_up(last, 1);
    if (lane == 0) recv = 0.0f;
    float diag = leftprev;  // dp[i+1][j-1]
    float left = recv;      // dp[i][j-1]
    float v = 0.0f;
    for (uint32_t c = 0; c < W; ++c) {
      const uint32_t j = lane * W + c;
      const float below = P[c * 64 + lane];  // dp[i+1][j]
      v = 0.0f;
      if (rowv && j < L && (int)j > i) {
        const uint32_t ui = (uint32_t)i;
        uint32_t t = 0;
        if (ui + 1 < j) { v = below; t = 1; }
        if (ui < j - 1 && v < left) { v = left; t = 2; }
        const uint32_t n = cc[j];
        if (ui + 1 < j - 1) {
          const float sc = Sb[c * 64 + lane];  // nussinov.cpp:236 / :329
          if (sc > 0.0f) {
            const float cand = diag + sc;
            ws.ck[(size_t)j * L + n] = ui;
            ws.cv[(size_t)j * L + n] = cand;
            cc[j] = n + 1;
            if (v < cand) { v = cand; t = 3; }
          }
        }
        for (uint32_t x = 0; x < n; ++x) {
          const uint32_t k = ws.ck[(size_t)j * L + x];
          const float dik = (k - 1 == ui) ? 0.0f : ld_l2(&ws.dp[(size_t)ui * L + k - 1]);  // dp[i][i] = 0 is never stored
          const float cand = dik + ws.cv[(size_t)j * L + x];
          if (v < cand) { v = cand; t = k - ui + 3; }
        }
        st_l2(&ws.dp[(size_t)ui * L + j], v);
        trb[tri_index(L, ui, j)] = (uint8_t)(t < 4 ? t : 4);
        if (t >= 4) trk[(size_t)ui * L + j] = t;
        if (ui == 0 && j == L - 1) score = v;
      }
      diag = below;
      P[c * 64 + lane] = v;
      left = v;
    }
    leftprev = recv;
    last = v;
#pragma unroll
    for (int c = 0; c < DD_WMAX; ++c)
      if ((uint32_t)c < W) Sb[c * 64 + lane] = nxt[c];
  }
  return __shfl(score, (int)((L - 1) / W));
}

// The same DP with everything on the per-cell dependency chain in LDS: `ring` holds the 64 rows in
// flight (row i at slot i & 63; a row is live for exactly the 64 steps its lanes sweep it), and the
// first DD_CAP candidates of every column sit in lck/lcv (the rest spill to ws.ck / ws.cv).
__device__ float nuss_wave_lds(uint32_t L, const float* __restrict__ S, const nuss_ws& ws, uint8_t* trb, uint32_t* trk, float* P, float* Sb,
                               uint32_t* cc, float* ring, uint32_t* lck, float* lcv, int lane) {
  const uint32_t W = (L + 63) / 64;
  for (uint32_t c = 0; c < W; ++c) {
    P[c * 64 + lane] = 0.0f;
    Sb[c * 64 + lane] = 0.0f;
    const uint32_t j = lane * W + c;
    if (j < L) cc[j] = 0;
  }
  float last = 0.0f, leftprev = 0.0f, score = 0.0f;
  const int nsteps = (int)L + 63;
  for (uint32_t c = 0; c < W; ++c) Sb[c * 64 + lane] = S[(size_t)c * 64 + lane];  // step 0
  for (int s = 0; s < nsteps; ++s) {
    const int i = (int)L - 1 - (s - lane);
    const bool rowv = i >= 0 && i < (int)L;
    // prefetch the scores of the next step (coalesced: S is stored in sweep order)
    float nxt[DD_WMAX];
    const bool nv = s + 1 < nsteps;
#pragma unroll
    for (int c = 0; c < DD_WMAX; ++c) nxt[c] = (nv && (uint32_t)c < W) ? S[((size_t)(s + 1) * W + c) * 64 + lane] : 0.0f;
    float recv = __shfl_up(last, 1);
    if (lane == 0) recv = 0.0f;
    float diag = leftprev;
    float left = recv;
    float v = 0.0f;
    float* rrow = ring + (size_t)((uint32_t)i & 63u) * L;
    for (uint32_t c = 0; c < W; ++c) {
      const uint32_t j = lane * W + c;
      const float below = P[c * 64 + lane];
      v = 0.0f;
      if (rowv && j < L && (int)j > i) {
        const uint32_t ui = (uint32_t)i;
        uint32_t t = 0;
        if (ui + 1 < j) { v = below; t = 1; }
        if (ui < j - 1 && v < left) { v = left; t = 2; }
        const uint32_t n = cc[j];
        if (ui + 1 < j - 1) {
          const float sc = Sb[c * 64 + lane];
          if (sc > 0.0f) {
            const float cand = diag + sc;
            if (n < DD_CAP) { lck[n * L + j] = ui; lcv[n * L + j] = cand; }
            else { ws.ck[(size_t)j * L + n] = ui; ws.cv[(size_t)j * L + n] = cand; }
            cc[j] = n + 1;
            if (v < cand) { v = cand; t = 3; }
          }
        }
        for (uint32_t x = 0; x < n; ++x) {
          uint32_t k; float cvv;
          if (x < DD_CAP) { k = lck[x * L + j]; cvv = lcv[x * L + j]; }
          else { k = ws.ck[(size_t)j * L + x]; cvv = ws.cv[(size_t)j * L + x]; }
          const float dik = (k - 1 == ui) ? 0.0f : rrow[k - 1];
          const float cand = dik + cvv;
          if (v < cand) { v = cand; t = k - ui + 3; }
        }
        rrow[j] = v;
        trb[tri_index(L, ui, j)] = (uint8_t)(t < 4 ? t : 4);
        if (t >= 4) trk[(size_t)ui * L + j] = t;
        if (ui == 0 && j == L - 1) score = v;
      }
      diag = below;
      P[c * 64 + lane] = v;
      left = v;
    }
    leftprev = recv;
    last = v;
#pragma unroll
    for (int c = 0; c < DD_WMAX; ++c)
      if ((uint32_t)c < W) Sb[c * 64 + lane] = nxt[c];
  }
  return __shfl(score, (int)((L - 1) / W));
}

// Register-resident forms for W <= DD_WREG columns per lane (W a template constant): the previous
// row, the scores and the candidate counters of the lane's columns live in registers, so a cell
// without candidates touches LDS only to publish its value; a cell with candidates makes two LDS
// round trips (all candidate keys/values at once, then all dp[i][k-1] at once).
#define DD_WREG 8
template <int W>
__device__ float nuss_wave_reg(uint32_t L, const float* __restrict__ S, const nuss_ws& ws, uint8_t* trb, uint32_t* trk, float* ring,
                               uint32_t* lck, float* lcv, int lane) {
  float P[W], Sc[W], nx[W];
  uint32_t n[W];
#pragma unroll
  for (int c = 0; c < W; ++c) { P[c] = 0.0f; n[c] = 0; Sc[c] = S[(size_t)c * 64 + lane]; nx[c] = 0.0f; }
  float last = 0.0f, leftprev = 0.0f, score = 0.0f;
  const int nsteps = (int)L + 63;
  for (int s = 0; s < nsteps; ++s) {
    const int i = (int)L - 1 - (s - lane);
    const bool rowv = i >= 0 && i < (int)L;
    if (s + 1 < nsteps) {
#pragma unroll
      for (int c = 0; c < W; ++c) nx[c] = S[((size_t)(s + 1) * W + c) * 64 + lane];
    }
    float recv = __shfl_up(last, 1);
    if (lane == 0) recv = 0.0f;
    float diag = leftprev;
    float left = recv;
    float v = 0.0f;
    const uint32_t ui = (uint32_t)i;
    float* rrow = ring + (size_t)(ui & 63u) * L;
#pragma unroll
    for (int c = 0; c < W; ++c) {
      const uint32_t j = lane * W + c;
      const float below = P[c];
      v = 0.0f;
      if (rowv && j < L && (int)j > i) {
        uint32_t t = 0;
        if (ui + 1 < j) { v = below; t = 1; }
        if (ui < j - 1 && v < left) { v = left; t = 2; }
        const uint32_t nc = n[c];
        if (ui + 1 < j - 1) {
          const float sc = Sc[c];
          if (sc > 0.0f) {
            const float cand = diag + sc;
            if (nc < DD_CAP) { lck[nc * L + j] = ui; lcv[nc * L + j] = cand; }
            else { ws.ck[(size_t)j * L + nc] = ui; ws.cv[(size_t)j * L + nc] = cand; }
            n[c] = nc + 1;
            if (v < cand) { v = cand; t = 3; }
          }
        }
        if (nc) {
          uint32_t kk[DD_CAP];
          float cvs[DD_CAP], dk[DD_CAP];
#pragma unroll
          for (int x = 0; x < DD_CAP; ++x) { kk[x] = lck[x * L + j]; cvs[x] = lcv[x * L + j]; }
#pragma unroll
          for (int x = 0; x < DD_CAP; ++x) dk[x] = rrow[(uint32_t)x < nc ? kk[x] - 1 : 0u];
#pragma unroll
          for (int x = 0; x < DD_CAP; ++x)
            if ((uint32_t)x < nc) {
              const float dik = (kk[x] - 1 == ui) ? 0.0f : dk[x];  // dp[i][i] = 0 is never stored
              const float cand = dik + cvs[x];
              if (v < cand) { v = cand; t = kk[x] - ui + 3; }
            }
          for (uint32_t x = DD_CAP; x < nc; ++x) {
            const uint32_t k = ws.ck[(size_t)j * L + x];
            const float dik = (k - 1 == ui) ? 0.0f : rrow[k - 1];
            const float cand = dik + ws.cv[(size_t)j * L + x];
            if (v < cand) { v = cand; t = k - ui + 3; }
          }
        }
        rrow[j] = v;
        trb[tri_index(L, ui, j)] = (uint8_t)(t < 4 ? t : 4);
        if (t >= 4) trk[(size_t)ui * L + j] = t;
        if (ui == 0 && j == L - 1) score = v;
      }
      diag = below;
      P[c] = v;
      left = v;
    }
    leftprev = recv;
    last = v;
#pragma unroll
    for (int c = 0; c < W; ++c) Sc[c] = nx[c];
  }
  return __shfl(score, (int)((L - 1) / W));
}

__device__ float nuss_wave_fast(uint32_t W, uint32_t L, const float* __restrict__ S, const nuss_ws& ws, uint8_t* trb, uint32_t* trk, float* ring,
                                uint32_t* lck, float* lcv, int lane) {
  switch (W) {
    case 1: return nuss_wave_reg<1>(L, S, ws, trb, trk, ring, lck, lcv, lane);
    case 2: return nuss_wave_reg<2>(L, S, ws, trb, trk, ring, lck, lcv, lane);
    case 3: return nuss_wave_reg<3>(L, S, ws, trb, trk, ring, lck, lcv, lane);
    case 4: return nuss_wave_reg<4>(L, S, ws, trb, trk, ring, lck, lcv, lane);
    case 5: return nuss_wave_reg<5>(L, S, ws, trb, trk, ring, lck, lcv, lane);
    case 6: return nuss_wave_reg<6>(L, S, ws, trb, trk, ring, lck, lcv, lane);
    case 7: return nuss_wave_reg<7>(L, S, ws, trb, trk, ring, lck, lcv, lane);
    default: return nuss_wave_reg<8>(L, S, ws, trb, trk, ring, lck, lcv, lane);
  }
}

template <int W>
__device__ float nw_wave_reg(uint32_t L1, uint32_t L2, const float* __restrict__ ps, const float* __restrict__ qs, float th,
                             const uint32_t* __restrict__ env, uint8_t* tr, int lane) {
  const uint32_t T = L2 + 1;
  float P[W], Pc[W], Qc[W], np[W], nq[W];
#pragma unroll
  for (int c = 0; c < W; ++c) { P[c] = 0.0f; Pc[c] = ps[(size_t)c * 64 + lane]; Qc[c] = qs[(size_t)c * 64 + lane]; np[c] = 0.0f; nq[c] = 0.0f; }
  float last = 0.0f, leftprev = 0.0f, score = 0.0f;
  const int nsteps = (int)L1 + 63;
  uint32_t ef = 1u, es = 0u;
  if (lane == 0) { ef = env[2]; es = env[3]; }  // row 1
  for (int s = 0; s < nsteps; ++s) {
    const int i = s - lane + 1;
    const bool rowv = i >= 1 && i <= (int)L1;
    if (s + 1 < nsteps) {
#pragma unroll
      for (int c = 0; c < W; ++c) { np[c] = ps[((size_t)(s + 1) * W + c) * 64 + lane]; nq[c] = qs[((size_t)(s + 1) * W + c) * 64 + lane]; }
    }
    const bool nrow = i + 1 >= 1 && i + 1 <= (int)L1;
    const uint32_t nef = nrow ? env[2 * (i + 1)] : 1u, nes = nrow ? env[2 * (i + 1) + 1] : 0u;
    const float recv = __shfl_up(last, 1);
    float diag = leftprev;
    float left = recv;
    float v = 0.0f;
#pragma unroll
    for (int c = 0; c < W; ++c) {
      const uint32_t k = lane * W + c;
      const float up = P[c];
      v = up;
      if (rowv && k <= L2) {
        if (k == 0) v = 0.0f;
        else if (k >= ef && k <= es) {
          v = diag + Pc[c] - th;
          v = v + Qc[c];
          uint8_t t = 'M';
          if (v < up) { v = up; t = 'X'; }
          if (v < left) { v = left; t = 'Y'; }
          tr[(size_t)i * T + k] = t;
        } else v = -FLT_MAX;
        if (i == (int)L1 && k == L2) score = v;
      }
      diag = up;
      P[c] = v;
      left = v;
    }
    leftprev = recv;
    last = v;
    ef = nef; es = nes;
#pragma unroll
    for (int c = 0; c < W; ++c) { Pc[c] = np[c]; Qc[c] = nq[c]; }
  }
  return __shfl(score, (int)(L2 / W));
}

__device__ float nw_wave_fast(uint32_t W, uint32_t L1, uint32_t L2, const float* __restrict__ ps, const float* __restrict__ qs, float th,
                              const uint32_t* __restrict__ env, uint8_t* tr, int lane) {
  switch (W) {
    case 1: return nw_wave_reg<1>(L1, L2, ps, qs, th, env, tr, lane);
    case 2: return nw_wave_reg<2>(L1, L2, ps, qs, th, env, tr, lane);
    case 3: return nw_wave_reg<3>(L1, L2, ps, qs, th, env, tr, lane);
    case 4: return nw_wave_reg<4>(L1, L2, ps, qs, th, env, tr, lane);
    case 5: return nw_wave_reg<5>(L1, L2, ps, qs, th, env, tr, lane);
    case 6: return nw_wave_reg<6>(L1, L2, ps, qs, th, env, tr, lane);
    case 7: return nw_wave_reg<7>(L1, L2, ps, qs, th, env, tr, lane);
    default: return nw_wave_reg<8>(L1, L2, ps, qs, th, env, tr, lane);
  }
}

// Sweep-order ("skewed") copies of the DP inputs: the value lane t needs at step s for its column c sits
// at ((s*W + c)*64 + t), so each step is one coalesced load per column.
__device__ __forceinline__ size_t nuss_skew(uint32_t L, uint32_t W, uint32_t i, uint32_t j) {
  const uint32_t lane = j / W, c = j - lane * W, step = L - 1 - i + lane;
  return ((size_t)step * W + c) * 64 + lane;
}
__device__ __forceinline__ size_t nw_skew(uint32_t W, uint32_t i, uint32_t k) {  // i in 1..L1, k in 0..L2
  const uint32_t lane = k / W, c = k - lane * W, step = i - 1 + lane;
  return ((size_t)step * W + c) * 64 + lane;
}
// all threads: S = w*(p-th)-q (nussinov.cpp:236); the association is the reference's
__device__ void dd_fill_scores(uint32_t L, const float* __restrict__ p, const float* __restrict__ q, float w, float th, float* S) {
  const uint32_t W = (L + 63) / 64;
  for (size_t c = threadIdx.x; c < (size_t)L * L; c += blockDim.x) {
    const uint32_t i = (uint32_t)(c / L), j = (uint32_t)(c - (size_t)i * L);
    S[nuss_skew(L, W, i, j)] = w * (p[c] - th) - q[c];
  }
}
__device__ void dd_fill_nw(uint32_t L1, uint32_t L2, const float* __restrict__ p, const float* __restrict__ q, float* ps, float* qs) {
  const uint32_t W = (L2 + 64) / 64;
  for (size_t c = threadIdx.x; c < (size_t)L1 * L2; c += blockDim.x) {
    const uint32_t i = (uint32_t)(c / L2), k = (uint32_t)(c - (size_t)i * L2);
    const size_t o = nw_skew(W, i + 1, k + 1);
    ps[o] = p[c];
    qs[o] = q[c];
  }
}

__device__ void nuss_traceback_b(uint32_t L, const uint8_t* trb, const uint32_t* trk, uint32_t* ss, uint32_t* stack) {
  // The segment being followed stays in registers; only the left half of a bifurcation is parked on
  // the stack (LDS, one packed word per segment, at most L/2 deep).  The pairs written do not depend
  // on the order the segments are visited in.
  uint32_t sp = 0;
  int i = 0, j = (int)L - 1;
  uint32_t guard = 4 * L + 8;
  while (guard--) {
    uint32_t t = 0;
    if (j > i) t = trb[tri_index(L, (uint32_t)i, (uint32_t)j)];  // tr of the diagonal is 0
    if (t == 0) {
      if (!sp) break;
      const uint32_t e = stack[--sp];
      i = (int)(e >> 16); j = (int)(e & 0xFFFFu);
      continue;
    }
    if (t == 1) ++i;
    else if (t == 2) --j;
    else if (t == 3) { ss[i] = j; ++i; --j; }
    else {
      const int k = i + (int)trk[(size_t)i * L + j] - 3;
      ss[k] = j;
      if (k - 1 > i) stack[sp++] = ((uint32_t)i << 16) | (uint32_t)(k - 1);
      i = k + 1; --j;
    }
  }
}

// SparseNeedlemanWunsch::decode DP (needleman_wunsch.cpp:276-296) by one wavefront; row i at step
// i-1+lane.  Cells outside the envelope hold lowest(), row 0 / column 0 hold 0.  tr must have been
// initialised by nw_init_tr.  Returns dp[L1][L2].
__device__ float nw_wave(uint32_t L1, uint32_t L2, const float* __restrict__ ps, const float* __restrict__ qs, float th,
                         const uint32_t* __restrict__ env, uint8_t* tr, float* P, float* Pb, float* Qb, int lane) {
  const uint32_t W = (L2 + 1 + 63) / 64, T = L2 + 1;
  for (uint32_t c = 0; c < W; ++c) {
    P[c * 64 + lane] = 0.0f;  // row 0
    Pb[c * 64 + lane] = ps[(size_t)c * 64 + lane];
    Qb[c * 64 + lane] = qs[(size_t)c * 64 + lane];
  }
  float last = 0.0f, leftprev = 0.0f, score = 0.0f;
  const int nsteps = (int)L1 + 63;
  uint32_t ef = 1u, es = 0u;
  if (lane == 0) { ef = env[2]; es = env[3]; }  // row 1
  for (int s = 0; s < nsteps; ++s) {
    const int i = s - lane + 1;
    const bool rowv = i >= 1 && i <= (int)L1;
    float np[DD_WMAX], nq[DD_WMAX];
    const bool nv = s + 1 < nsteps;
#pragma unroll
    for (int c = 0; c < DD_WMAX; ++c) {
      const bool on = nv && (uint32_t)c < W;
      np[c] = on ? ps[((size_t)(s + 1) * W + c) * 64 + lane] : 0.0f;
      nq[c] = on ? qs[((size_t)(s + 1) * W + c) * 64 + lane] : 0.0f;
    }
    const bool nrow = i + 1 >= 1 && i + 1 <= (int)L1;
    const uint32_t nef = nrow ? env[2 * (i + 1)] : 1u, nes = nrow ? env[2 * (i + 1) + 1] : 0u;
    const float recv = __shfl_up(last, 1);
    float diag = leftprev;  // dp[i-1][k-1]
    float left = recv;      // dp[i][k-1]
    float v = 0.0f;
    for (uint32_t c = 0; c < W; ++c) {
      const uint32_t k = lane * W + c;
      const float up = P[c * 64 + lane];  // dp[i-1][k]
      v = up;  // rows not started yet keep row 0
      if (rowv && k <= L2) {
        if (k == 0) v = 0.0f;
        else if (k >= ef && k <= es) {
          v = diag + Pb[c * 64 + lane] - th;
          v = v + Qb[c * 64 + lane];
          uint8_t t = 'M';
          if (v < up) { v = up; t = 'X'; }
          if (v < left) { v = left; t = 'Y'; }
          tr[(size_t)i * T + k] = t;
        } else v = -FLT_MAX;
        if (i == (int)L1 && k == L2) score = v;
      }
      diag = up;
      P[c * 64 + lane] = v;
      left = v;
    }
    leftprev = recv;
    last = v;
    ef = nef; es = nes;
#pragma unroll
    for (int c = 0; c < DD_WMAX; ++c)
      if ((uint32_t)c < W) { Pb[c * 64 + lane] = np[c]; Qb[c * 64 + lane] = nq[c]; }
  }
  return __shfl(score, (int)(L2 / W));
}

__device__ void nw_init_tr(uint32_t L1, uint32_t L2, uint8_t* tr) {  // needleman_wunsch.cpp:264-274
  const uint32_t T = L2 + 1;
  for (size_t c = threadIdx.x; c < (size_t)(L1 + 1) * T; c += blockDim.x) {
    const uint32_t i = (uint32_t)(c / T), k = (uint32_t)(c % T);
    tr[c] = (i == 0 && k == 0) ? ' ' : (k == 0 ? 'X' : (i == 0 ? 'Y' : ' '));
  }
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
  __shared__ float s_score[3];
  float c = 0.0f, eta = prm.eta0, s_prev = 0.0f;  // meaningful in thread 0
  uint32_t t = 0, violated = 0;
  if (tid == 0) { s_eta = eta; s_bad = 0; }
  // dynamic LDS: previous-row buffers and candidate counters of the three wave DPs, then whichever
  // traceback tables fit (nd.lds_flags, decided by the host): bit 0 alignment, bit 1 x, bit 2 y
  extern __shared__ unsigned char s_dd[];
  const uint32_t Wx = (L1 + 63) / 64, Wy = (L2 + 63) / 64, Wz = (L2 + 64) / 64;
  float* Px = (float*)s_dd;
  float* Sbx = Px + Wx * 64;
  float* Py = Sbx + Wx * 64;
  float* Sby = Py + Wy * 64;
  float* Pz = Sby + Wy * 64;
  float* Pbz = Pz + Wz * 64;
  float* Qbz = Pbz + Wz * 64;
  uint32_t* ccx = (uint32_t*)(Qbz + Wz * 64);
  uint32_t* ccy = ccx + L1;
  unsigned char* lds_tail = (unsigned char*)(ccy + L2);
  const size_t nz = (size_t)(L1 + 1) * (L2 + 1), nx = (size_t)L1 * (L1 + 1) / 2, ny = (size_t)L2 * (L2 + 1) / 2;
  uint8_t* trz = nd.tr_z;
  uint8_t* trx = nd.trb_x;
  uint8_t* try_ = nd.trb_y;
  if (nd.lds_flags & 1) { trz = lds_tail; lds_tail += (nz + 15) & ~(size_t)15; }
  if (nd.lds_flags & 2) { trx = lds_tail; lds_tail += (nx + 15) & ~(size_t)15; }
  if (nd.lds_flags & 4) { try_ = lds_tail; lds_tail += (ny + 15) & ~(size_t)15; }
  float *ringx = nullptr, *ringy = nullptr, *lcvx = nullptr, *lcvy = nullptr;
  uint32_t *lckx = nullptr, *lcky = nullptr;
  if (nd.lds_flags & 8) { ringx = (float*)lds_tail; lckx = (uint32_t*)(ringx + 64 * (size_t)L1); lcvx = (float*)(lckx + DD_CAP * (size_t)L1); lds_tail = (unsigned char*)(lcvx + DD_CAP * (size_t)L1); }
  if (nd.lds_flags & 16) { ringy = (float*)lds_tail; lcky = (uint32_t*)(ringy + 64 * (size_t)L2); lcvy = (float*)(lcky + DD_CAP * (size_t)L2); lds_tail = (unsigned char*)(lcvy + DD_CAP * (size_t)L2); }
  nw_init_tr(L1, L2, trz);
  // sweep-order inputs of the three DPs, built once; the multiplier updates below keep them current
  dd_fill_scores(L1, nd.p_x, nd.q_x, w_x, prm.th_s, nd.s_x);
  dd_fill_scores(L2, nd.p_y, nd.q_y, w_y, prm.th_s, nd.s_y);
  dd_fill_nw(L1, L2, nd.p_z, nd.q_z, nd.pz_s, nd.qz_s);
  __syncthreads();
  const int wave = (int)(tid >> 6), lane = (int)(tid & 63);

  // optional phase timing (100 MHz ticks accumulated over the iterations into info[8..13]; tuning aid)
  unsigned long long tk[7] = {0, 0, 0, 0, 0, 0, 0}, t_prev = 0;
#define DD_TICK(k) if (prm.stamps && tid == 0) { const unsigned long long now = wall_clock64(); tk[k] += now - t_prev; t_prev = now; }
  if (prm.stamps && tid == 0) t_prev = wall_clock64();
  for (t = 0; t != prm.t_max; ++t) {
    for (uint32_t i = tid; i < L1; i += nt) nd.x[i] = DD_NONE;
    for (uint32_t k = tid; k < L2; k += nt) nd.y[k] = DD_NONE;
    // the three subproblems (dafs.cpp:1091-1093) side by side, one wavefront each, DP then traceback
    __syncthreads();
    if (wave == 0) {
      const float sc = (ringx && Wx <= DD_WREG) ? nuss_wave_fast(Wx, L1, nd.s_x, nd.wx, trx, nd.trk_x, ringx, lckx, lcvx, lane)
                       : ringx ? nuss_wave_lds(L1, nd.s_x, nd.wx, trx, nd.trk_x, Px, Sbx, ccx, ringx, lckx, lcvx, lane)
                             : nuss_wave(L1, nd.s_x, nd.wx, trx, nd.trk_x, Px, Sbx, ccx, lane);
      DD_TICK(0);
      if (lane == 0) { s_score[0] = sc; nuss_traceback_b(L1, trx, nd.trk_x, nd.x, (uint32_t*)Px); }
      DD_TICK(1);
    } else if (wave == 1) {
      const float sc = (ringy && Wy <= DD_WREG) ? nuss_wave_fast(Wy, L2, nd.s_y, nd.wy, try_, nd.trk_y, ringy, lcky, lcvy, lane)
                       : ringy ? nuss_wave_lds(L2, nd.s_y, nd.wy, try_, nd.trk_y, Py, Sby, ccy, ringy, lcky, lcvy, lane)
                             : nuss_wave(L2, nd.s_y, nd.wy, try_, nd.trk_y, Py, Sby, ccy, lane);
      if (lane == 0) { s_score[1] = sc; nuss_traceback_b(L2, try_, nd.trk_y, nd.y, (uint32_t*)Py); }
    } else if (wave == 2) {
      const float sc = Wz <= DD_WREG ? nw_wave_fast(Wz, L1, L2, nd.pz_s, nd.qz_s, prm.th_a, nd.env, trz, lane)
                                     : nw_wave(L1, L2, nd.pz_s, nd.qz_s, prm.th_a, nd.env, trz, Pz, Pbz, Qbz, lane);
      if (lane == 0) { s_score[2] = sc; if (!nw_traceback(L1, L2, trz, nd.z)) s_bad = 1; }
    }
    DD_TICK(0);
    if (tid >= 192) {
      for (uint32_t e = tid - 192; e < npx; e += nt - 192) nd.tx[e] = 0;
      for (uint32_t e = tid - 192; e < npy; e += nt - 192) nd.ty[e] = 0;
      for (uint32_t e = tid - 192; e < ncz; e += nt - 192) nd.tz[e] = 0;
    }
    if (tid == 0) s_violated = 0;
    __syncthreads();
    DD_TICK(2);

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
    DD_TICK(3);
    eta = s_eta;

    // multiplier updates (:1121-1254), every cell touched by exactly one lane
    uint32_t viol = 0;
    for (uint32_t i = tid; i < L1; i += nt) {
      const uint32_t j = nd.x[i];
      if (j != DD_NONE) {
        const int32_t id = nd.xmap[(size_t)i * L1 + j];
        const int tc = id >= 0 ? nd.tx[id] : 0;
        if (tc != 1) {
          ++viol;
          const size_t o = (size_t)i * L1 + j;
          const float qn = nd.q_x[o] - eta * (tc - 1);
          nd.q_x[o] = qn;
          nd.s_x[nuss_skew(L1, Wx, i, j)] = w_x * (nd.p_x[o] - prm.th_s) - qn;
        }
      }
      for (uint32_t e = nd.px_ptr[i]; e < nd.px_ptr[i + 1]; ++e) {
        if (!nd.cx_flag[e]) continue;
        const uint32_t jj = nd.px_j[e];
        const int tc = nd.tx[e];
        if (j != jj && tc != 0) {
          ++viol;
          const size_t o = (size_t)i * L1 + jj;
          const float qn = nd.q_x[o] - eta * tc;
          nd.q_x[o] = qn;
          nd.s_x[nuss_skew(L1, Wx, i, jj)] = w_x * (nd.p_x[o] - prm.th_s) - qn;
        }
      }
      const uint32_t kz = nd.z[i];
      if (kz != DD_NONE) {
        const int32_t id = nd.zmap[(size_t)i * L2 + kz];
        const int tc = id >= 0 ? nd.tz[id] : 0;
        if (tc > 1) ++viol;
        const float v = nd.q_z[(size_t)i * L2 + kz] - eta * (1 - tc);
        const float qn = (0.0f < v) ? v : 0.0f;
        nd.q_z[(size_t)i * L2 + kz] = qn;
        nd.qz_s[nw_skew(Wz, i + 1, kz + 1)] = qn;
      }
      for (uint32_t e = nd.cz_ptr[i]; e < nd.cz_ptr[i + 1]; ++e) {
        const uint32_t kk = nd.cz_k[e];
        if (kz != kk) {
          const int tc = nd.tz[e];
          if (tc > 0) ++viol;
          const float v = nd.q_z[(size_t)i * L2 + kk] + eta * tc;
          const float qn = (0.0f < v) ? v : 0.0f;
          nd.q_z[(size_t)i * L2 + kk] = qn;
          nd.qz_s[nw_skew(Wz, i + 1, kk + 1)] = qn;
        }
      }
    }
    for (uint32_t k = tid; k < L2; k += nt) {
      const uint32_t l = nd.y[k];
      if (l != DD_NONE) {
        const int32_t id = nd.ymap[(size_t)k * L2 + l];
        const int tc = id >= 0 ? nd.ty[id] : 0;
        if (tc != 1) {
          ++viol;
          const size_t o = (size_t)k * L2 + l;
          const float qn = nd.q_y[o] - eta * (tc - 1);
          nd.q_y[o] = qn;
          nd.s_y[nuss_skew(L2, Wy, k, l)] = w_y * (nd.p_y[o] - prm.th_s) - qn;
        }
      }
      for (uint32_t e = nd.py_ptr[k]; e < nd.py_ptr[k + 1]; ++e) {
        if (!nd.cy_flag[e]) continue;
        const uint32_t ll = nd.py_l[e];
        const int tc = nd.ty[e];
        if (l != ll && tc != 0) {
          ++viol;
          const size_t o = (size_t)k * L2 + ll;
          const float qn = nd.q_y[o] - eta * tc;
          nd.q_y[o] = qn;
          nd.s_y[nuss_skew(L2, Wy, k, ll)] = w_y * (nd.p_y[o] - prm.th_s) - qn;
        }
      }
    }
    if (viol) atomicAdd(&s_violated, viol);
    __syncthreads();
    DD_TICK(4);

    if (tid == 0) {
      // dual value in the reference's summation order (:1090-1093, :1111)
      float s = 0.0f;
      s += s_score[0];
      s += s_score[1];
      s += s_score[2];
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
    DD_TICK(5);
    if (s_stop) break;
  }
  if (tid == 0) {
    *nd.score = s_prev;
    nd.info[1] = t;
    nd.info[2] = violated;
    nd.info[3] = s_bad ? 1u : 0u;
    if (prm.stamps)
      for (int k = 0; k < 6; ++k) nd.info[8 + k] = (uint32_t)tk[k];
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
int dd_solve_launch(const dd_node* d_nodes, uint32_t nnodes, dd_params prm, size_t lds_bytes, hipStream_t st) {
  if (!nnodes) return DAFS_HIP_OK;
  static bool attr = false;
  if (!attr) {
    if (hip_check(hipFuncSetAttribute((const void*)k_dd_solve, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kDdLdsBudget))) return DAFS_HIP_ELAUNCH;
    attr = true;
  }
  if (lds_bytes > kDdLdsBudget) return DAFS_HIP_EINVAL;
  hipLaunchKernelGGL(k_dd_solve, dim3(nnodes), dim3(DD_THREADS), lds_bytes, st, d_nodes, prm);
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
