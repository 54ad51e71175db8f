// dafs_amd/csrc/dd.hip -- the progressive phase on the GPU: posterior averaging, the consensus
// base-pair structure, and the dual-decomposition loop with its three DP subproblems.
//
// Reference functions replaced (paths relative to /root/reference):
//   average_basepairing_probability / average_matching_probability   src/dafs.cpp:513-607
//   DAFS::solve_by_dd (cbp enumeration, multiplier updates, step size) src/dafs.cpp:1006-1295
//   SparseNussinov::decode (both overloads)                           src/nussinov.cpp:207-392
//   SparseNeedlemanWunsch::initialize / decode (both overloads)       src/needleman_wunsch.cpp:198-422
//
// One workgroup solves one guide-tree node (in split mode three: a leader and one per folding DP);
// independent nodes of a launch run side by side.  The subgradient loop stays on the device and is
// resumable: a launch runs at most prm.slice iterations of every node, parks the loop state of the
// unfinished ones and returns, so the host can merge finished nodes and open their parents without a
// level barrier (capi_dd.cpp: dafs_hip_nodes_*).  Per iteration the three subproblems run on three
// wavefronts as skewed, register-resident DPs (lane t owns W columns; the folding DPs sweep rows from the
// bottom, the alignment DP from the top) with their inputs stored in sweep order, the rows in flight,
// candidate split points and traceback codes in LDS (the codes in HBM from ~415 columns on; 48 lanes of up
// to 16 columns from 513 to 768), followed by their tracebacks (the folding one by the whole wavefront, run
// by run); the multiplier update is parallel over the sparse consensus structure.
// A folding DP that has no register form (beyond 1024 columns) runs span-ordered on all threads of its folder's
// workgroup with its rows in flight in LDS (nuss_wg_span); one whose register form overflows its candidate slots falls
// back to the span-ordered form on global tables (nuss_pair_dp).  The alignment DP stays in registers at any width: beyond
// 16 columns per lane its traceback codes go to HBM slots, beyond 2047 columns it runs in column panels (nw_wave_reg).
// Float sums that the reference forms sequentially (the dual value s, which steers the step size) are formed in the
// same order: positive terms are compacted in consensus-pair order and added by one lane.
// The standalone decoders (the plugin entry points and the final consensus structure): k_nussinov_single takes the
// workgroup form up to ~9 900 columns, k_nw_single the barrier-per-diagonal form below.
#include <hip/hip_runtime.h>
#include <float.h>
#include <stdint.h>
#include <type_traits>
#include "../../include/dafs_hip.h"
#include "dd.h"
#include "hip_util.h"
#include "stage.h"

namespace dafs {

// Address-space-qualified views: the wave DPs must compile to ds_* / global_* instructions, not flat_* ones (a flat
// access counts on both wait counters, so an LDS operand would wait for every global store and load in flight)
#define DD_LDS __attribute__((address_space(3)))
#define DD_GLB __attribute__((address_space(1)))
#define DD_CUTOFF 0.01f  // reference CUTOFF (double 0.01): for float v, v > 0.01 <=> v > 0.01f
#define DD_NONE 0xFFFFFFFFu
#define DD_THREADS 512
// k_dd_solve: four wavefronts (x, y, z, housekeeping), one per SIMD, so each may use the whole register file
#define DD_SOLVE_THREADS 256

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
// The workgroup form of the folding DP for alignments beyond the register forms (more than DD_WFOLD columns per lane, i.e.
// more than 1024 columns), nussinov.cpp:207-298 again.  Span-ordered like nuss_pair_dp, but nothing that the next span
// waits for crosses global memory unless a column has candidates:
//   * a cell's three neighbours dp[i+1][j], dp[i][j-1], dp[i+1][j-1] are the two previous spans: three rolling rows in LDS;
//   * its score comes from the by-span copy S[(j-i)*Lp + i] (dd_fill_scores, kept current by the multiplier updates), one
//     coalesced load per span, fetched a span ahead for the first cells of every thread;
//   * the candidate counters and the first K candidates {k, dp[k+1][j-1] + s_kj} of every column sit in LDS, later ones in the
//     global lists of nuss_ws; the bifurcation terms dp[i][k-1] are the only gathers from the table, issued together for the
//     four cells a thread handles at a time;
//   * dp and the traceback codes are written by span, D[(j-i)*L + i] (the arrays of nuss_ws, re-indexed): coalesced stores that
//     nobody waits for -- a bifurcation reads dp[i][k-1] with k <= j-3, a cell written at least four barriers earlier.
// Same cells, same comparisons in the same order as nuss_cell, so the same table, codes and structure.
// DIRECT (the standalone decoder, which has no by-span copy): the score comes from p (and q) as nuss_cell computes it, a
// strided read per cell, still a span ahead.
template <int K, bool DIRECT = false>
__device__ float nuss_wg_span(uint32_t L, const float* __restrict__ S_, const nuss_ws& ws, float* lds, const float* __restrict__ q_ = nullptr,
                              float w = 0.0f, float th = 0.0f) {
  const uint32_t tid = threadIdx.x, nt = blockDim.x;
  const uint32_t Lr = (L + 3) & ~3u, Lp = (L + 63) & ~63u;
  // address-space-qualified views (see DD_LDS): a flat access to LDS would wait for every global store in flight
  DD_LDS float* buf = (DD_LDS float*)lds;
  DD_LDS uint32_t* cc = (DD_LDS uint32_t*)(buf + 3 * Lr);
  DD_LDS uint32_t* hk = cc + Lr;
  DD_LDS float* hv = (DD_LDS float*)(hk + (K ? K : 1) * Lr);
  DD_GLB const float* S = (DD_GLB const float*)S_;
  DD_GLB const float* Q = (DD_GLB const float*)q_;
  auto score = [&](uint32_t l, uint32_t i) -> float {  // s of cell (i, i + l); 0 where the reference's span test fails
    if constexpr (DIRECT) {
      if (l < 3) return 0.0f;
      const size_t o = (size_t)i * L + i + l;
      return Q ? w * (S[o] - th) - Q[o] : S[o] - th;  // nussinov.cpp:236 / :329
    } else return S[(size_t)l * Lp + i];
  };
  DD_GLB float* D = (DD_GLB float*)ws.dp;
  DD_GLB uint32_t* T = (DD_GLB uint32_t*)ws.tr;
  DD_GLB uint32_t* gck = (DD_GLB uint32_t*)ws.ck;
  DD_GLB float* gcv = (DD_GLB float*)ws.cv;
  for (uint32_t i = tid; i < L; i += nt) {
    buf[i] = 0.0f; buf[Lr + i] = 0.0f; buf[2 * Lr + i] = 0.0f; cc[i] = 0;
    D[i] = 0.0f; T[i] = 0;                                  // span 0
    if (i + 1 < L) { D[(size_t)L + i] = 0.0f; T[(size_t)L + i] = 0; }  // span 1: neither neighbour test of nuss_cell holds
  }
  constexpr int U = 4;  // cells of a thread whose loads are in flight together
  float s_ahead[U];
#pragma unroll
  for (int u = 0; u < U; ++u) { const uint32_t i = tid + u * nt; s_ahead[u] = (2 < L && i < L - 2) ? score(2, i) : 0.0f; }
  __syncthreads();
  for (uint32_t l = 2; l < L; ++l) {
    DD_LDS const float* p1 = buf + ((l - 1) % 3) * Lr;
    DD_LDS const float* p2 = buf + ((l - 2) % 3) * Lr;
    DD_LDS float* cur = buf + (l % 3) * Lr;
    const uint32_t ncell = L - l;
    for (uint32_t base = 0; base < ncell; base += U * nt) {
      float sc[U], g[U][K ? K : 1], hvx[U][K ? K : 1];
      uint32_t kk[U][K ? K : 1], n[U];
      // Everything a cell reads from memory, in three batches without a branch between them (as separate conditional blocks
      // every candidate cost the cell a dependent LDS round trip of its own): the counters, then all K heads of all U cells
      // whether they are filled or not, then the gathers with the unfilled ones pointed at dp[i][i] -- one LDS trip, one
      // more, one trip to the table per span.
      uint32_t ci[U];  // the cell's row, clamped into the span for the lanes beyond it (they read, and discard)
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint32_t i = base + u * nt + tid;
        ci[u] = i < ncell ? i : ncell - 1;
        n[u] = cc[ci[u] + l];
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint32_t i = base + u * nt + tid;
        sc[u] = i < ncell ? (base == 0 ? s_ahead[u] : score(l, i)) : 0.0f;
#pragma unroll
        for (int x = 0; x < K; ++x) { kk[u][x] = hk[x * Lr + ci[u] + l]; hvx[u][x] = hv[x * Lr + ci[u] + l]; }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (base + u * nt + tid >= ncell) n[u] = 0;
#pragma unroll
        for (int x = 0; x < K; ++x) {
          const uint32_t d = (uint32_t)x < n[u] ? kk[u][x] - 1 - ci[u] : 0u;  // span of dp[i][k-1]; an unfilled head reads dp[i][i]
          g[u][x] = D[(size_t)(d < l ? d : 0u) * L + ci[u]];
        }
      }
      if (base == 0 && l + 1 < L) {
#pragma unroll
        for (int u = 0; u < U; ++u) { const uint32_t i = tid + u * nt; s_ahead[u] = i < ncell - 1 ? score(l + 1, i) : 0.0f; }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint32_t i = base + u * nt + tid, j = i + l;
        if (i >= ncell) continue;
        float v = p1[i + 1];
        uint32_t t = 1;
        { const float b = p1[i]; if (v < b) { v = b; t = 2; } }
        if (sc[u] > 0.0f) {  // S holds 0 where j - i < 3 (the reference's span test)
          const float c = p2[i + 1] + sc[u];
          if (n[u] < (uint32_t)K) { hk[n[u] * Lr + j] = i; hv[n[u] * Lr + j] = c; }
          else { gck[(size_t)j * L + n[u]] = i; gcv[(size_t)j * L + n[u]] = c; }
          cc[j] = n[u] + 1;
          if (v < c) { v = c; t = 3; }
        }
#pragma unroll
        for (int x = 0; x < K; ++x)
          if ((uint32_t)x < n[u]) { const float c = g[u][x] + hvx[u][x]; if (v < c) { v = c; t = kk[u][x] - i + 3; } }
        for (uint32_t x = K; x < n[u]; ++x) {
          const uint32_t k = gck[(size_t)j * L + x];
          const float c = D[(size_t)(k - 1 - i) * L + i] + gcv[(size_t)j * L + x];
          if (v < c) { v = c; t = k - i + 3; }
        }
        cur[i] = v;
        D[(size_t)l * L + i] = v;
        T[(size_t)l * L + i] = t;
      }
    }
    __syncthreads();
  }
  return buf[((L - 1) % 3) * Lr];
}

// The traceback over the by-span codes of nuss_wg_span, by one wavefront (cf. nuss_traceback_fast): the walk consists of
// runs -- stretches of code 1 (i+1), of code 2 (j-1) and of stacked pairs (code 3) -- and the lanes read the next 64 cells
// along the current direction at once; a ballot finds where the run ends and the lane that found it already holds the code
// of the cell the walk lands on: one trip to the table per run instead of one per cell.  A bifurcation (code k - i + 3)
// pairs (k, j), parks (i, k-1) on the stack (LDS, 16 + 16 bits) and goes on with (k+1, j-1).
__device__ void nuss_traceback_span(uint32_t L, const uint32_t* __restrict__ T_, uint32_t* ss_, uint32_t* stack_, int lane) {
  DD_GLB const uint32_t* T = (DD_GLB const uint32_t*)T_;
  DD_LDS uint32_t* stack = (DD_LDS uint32_t*)stack_;
  DD_GLB uint32_t* ss = (DD_GLB uint32_t*)ss_;
  auto code = [&](int i, int j) -> uint32_t { return (i >= 0 && j > i && j < (int)L) ? T[(size_t)(j - i) * L + i] : 0u; };
  uint32_t sp = 0;
  int i = 0, j = (int)L - 1;
  uint32_t guard = 4 * L + 8;
  uint32_t t = (uint32_t)__builtin_amdgcn_readfirstlane((int)code(i, j));
  while (guard--) {
    if (t == 0) {
      if (!sp) break;
      --sp;
      const uint32_t e = (uint32_t)__builtin_amdgcn_readfirstlane((int)stack[sp]);
      i = (int)(e >> 16); j = (int)(e & 0xFFFFu);
      t = (uint32_t)__builtin_amdgcn_readfirstlane((int)code(i, j));
    } else if (t == 1) {
      const uint32_t probe = code(i + 1 + lane, j);
      const unsigned long long m = __ballot(probe != 1u);
      const int r = m ? (int)__ffsll((long long)m) - 1 : 63;
      i += 1 + r;
      t = (uint32_t)__builtin_amdgcn_readlane((int)probe, r);
    } else if (t == 2) {
      const uint32_t probe = code(i, j - 1 - lane);
      const unsigned long long m = __ballot(probe != 2u);
      const int r = m ? (int)__ffsll((long long)m) - 1 : 63;
      j -= 1 + r;
      t = (uint32_t)__builtin_amdgcn_readlane((int)probe, r);
    } else if (t == 3) {
      const uint32_t probe = code(i + 1 + lane, j - 1 - lane);
      const unsigned long long m = __ballot(probe != 3u);
      const int r = m ? (int)__ffsll((long long)m) - 1 : 63;
      if (lane == 0) ss[i] = (uint32_t)j;
      if (lane < r) ss[i + 1 + lane] = (uint32_t)(j - 1 - lane);
      i += 1 + r; j -= 1 + r;
      t = (uint32_t)__builtin_amdgcn_readlane((int)probe, r);
    } else {
      const int k = i + (int)t - 3;
      if (lane == 0) {
        ss[k] = (uint32_t)j;
        if (k - 1 > i) stack[sp] = ((uint32_t)i << 16) | (uint32_t)(k - 1);
      }
      if (k - 1 > i) ++sp;
      wave_lds_fence();
      i = k + 1; --j;
      t = (uint32_t)__builtin_amdgcn_readfirstlane((int)code(i, j));
    }
  }
}

// ------------------------------------------------------------------------------------------
// SparseNeedlemanWunsch
// ------------------------------------------------------------------------------------------
// initialize (needleman_wunsch.cpp:198-253); fa/la = scratch of L1+1 uint32 each
__device__ void nw_envelope(uint32_t L1, uint32_t L2, const float* __restrict__ p, float th, uint32_t* env,
                            uint32_t* fa, uint32_t* la) {
  // first / last column of every row with p - th >= 0 in parallel, then the sequential smoothing passes
  // (needleman_wunsch.cpp:196-243) by one thread on an LDS copy (2 x 1025 words) instead of global memory
  __shared__ uint32_t s_env[2 * 1025], s_fl[2 * 1025];
  const uint32_t tid = threadIdx.x, nt = blockDim.x;
  const bool in_lds = L1 <= 1024;
  if (in_lds) { fa = s_fl; la = s_fl + 1025; }
  {  // a wavefront per row: one pass over the row, first and last column with p - th >= 0 from the ballots
    const uint32_t wave = tid >> 6, lane = tid & 63, nwaves = nt >> 6;
    for (uint32_t i = wave + 1; i <= L1; i += nwaves) {
      uint32_t f = 0, l = 0;
      for (uint32_t k0 = 0; k0 < L2; k0 += 64) {
        const uint32_t k = k0 + lane;  // column k+1 of the reference's 1-based walk
        const unsigned long long m = __ballot(k < L2 && p[(size_t)(i - 1) * L2 + k] - th >= 0.0f);
        if (m) {
          if (!f) f = k0 + (uint32_t)__ffsll((long long)m);
          l = k0 + 64 - (uint32_t)__clzll((long long)m);
        }
      }
      if (lane == 0) { fa[i] = f; la[i] = l; }
    }
  }
  __syncthreads();
  uint32_t* ev = in_lds ? s_env : env;
  if (tid == 0) {
    for (uint32_t i = 0; i <= L1; ++i) { ev[2 * i] = 0; ev[2 * i + 1] = 0; }
    for (uint32_t i = 1; i <= L1; ++i) {
      const uint32_t f = fa[i], l = la[i];
      if (f) {
        if (f - 1 < ev[2 * (i - 1)]) ev[2 * (i - 1)] = f - 1;
        ev[2 * i] = f;
      }
      if (ev[2 * i] == 0) {
        ev[2 * i] = ev[2 * (i - 1)];
        ev[2 * i + 1] = ev[2 * (i - 1) + 1];
        continue;
      }
      if (l - 1 > ev[2 * (i - 1) + 1]) ev[2 * (i - 1) + 1] = l - 1;
      ev[2 * i + 1] = l;
    }
    ev[2 * L1 + 1] = L2;
    for (uint32_t i = L1, v = L2; i != 0; --i) { v = v < ev[2 * i] ? v : ev[2 * i]; ev[2 * i] = v; }
    for (uint32_t i = 0, v = 0; i != L1 + 1; ++i) { v = v > ev[2 * i + 1] ? v : ev[2 * i + 1]; ev[2 * i + 1] = v; }
    for (uint32_t i = 1; i != L1 + 1; ++i)
      if (ev[2 * (i - 1) + 1] < ev[2 * i]) ev[2 * i] = ev[2 * (i - 1) + 1];
  }
  __syncthreads();
  if (in_lds) {
    for (uint32_t i = tid; i < 2 * (L1 + 1); i += nt) env[i] = s_env[i];
    __syncthreads();
  }
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
__device__ __forceinline__ float ld_l2g(DD_GLB const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_l2g(DD_GLB float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_l2(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ size_t tri_index(uint32_t L, uint32_t i, uint32_t j) { return (size_t)i * L - (size_t)i * (i - 1) / 2 + (j - i); }  // j >= i


// Register-resident forms (W columns per lane, a template constant: up to DD_WNW for the alignment DP, DD_WREG for
// the folding DP with its codes in LDS, up to DD_WFOLD for the folding DP with its codes in HBM): the previous
// row, the scores and the candidates of the lane's columns live in registers; the only LDS traffic of a cell
// is publishing its value and code and reading the dp[i][k-1] of its column's candidates (fetched a step
// ahead for W <= 4).
// Address-space-qualified views: the loops below must compile to ds_* / global_* instructions, not
// flat_* ones (a flat access waits on both counters, i.e. on the prefetch of the next step as well).

// lane t receives lane t-1's value, lane 0 receives 0: one DPP move (wave_shr:1) instead of a trip through the
// LDS crossbar (ds_bpermute), which sits on the step-to-step dependency chain of the wave DPs
__device__ __forceinline__ float wave_shr1(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xf, 0xf, false));
}
// The loop must not contain a global store either: on gfx9 loads and stores share vmcnt and complete
// out of order with each other, so one possible store in flight turns every wait into vmcnt(0) and
// the prefetch of the next step is waited for at once.  Hence: traceback codes in LDS (a bifurcation
// is recorded as 4 + the index of its candidate, whose split row is read back from lck), and a column that
// collects more than DD_CAP candidates raises `ovf`; the caller then repeats the DP with
// nuss_pair_dp, the span-ordered form in global memory.
// TRG: the traceback codes go to HBM instead (one byte per cell of the upper triangle, trbg_) for alignments whose
// nibble table no longer fits beside the rows in flight.  That puts stores into the loop, so every wait for the
// prefetched scores also waits for them; with the seven or more cells per lane of such alignments a step is
// longer than the trip to L2 and the wait is over before it starts.
template <int W, bool TRG>
__device__ float nuss_wave_reg(uint32_t L, const float* S_, uint32_t* trb_, uint8_t* trbg_, float* ring_, uint32_t* lck_, int lane, bool* ovf_out) {
  DD_GLB uint8_t* trbg = (DD_GLB uint8_t*)trbg_;
  const uint32_t R = dd_ring_rows(L);  // rows in flight = lanes at work; row i lives in slot i mod R
  DD_GLB const float* S = (DD_GLB const float*)S_;
  DD_LDS char* ring = (DD_LDS char*)ring_;
  DD_LDS uint32_t* lck = (DD_LDS uint32_t*)lck_;
  DD_LDS uint32_t* trb = (DD_LDS uint32_t*)trb_;  // one nibble per cell of the upper triangle, zeroed by the caller
  // Per owned column: previous row, score, and the column's candidates in the order they were found (slot x =
  // x-th candidate; empty slots hold -inf so they never win): value dp[k+1][j-1]+s and the byte offset of
  // dp[.][k-1] within a row of the ring.  A single wavefront issues one instruction every four cycles whatever
  // its kind, so the step is as fast as it is short: the cell is straight-line selects (the lanes are on
  // different rows and columns, a branch would be taken by somebody anyway) except for the rare insertion of a
  // new candidate; S holds 0 where j - i < 3 (dd_fill_scores), which stands in for the reference's span test.
  float P[W], Sc[W], nx[W], cvs[W][DD_CAP];
  uint32_t koff[W][DD_CAP], n[W];
#pragma unroll
  for (int c = 0; c < W; ++c) {
    P[c] = 0.0f; n[c] = 0; Sc[c] = S[(size_t)c * 64 + lane]; nx[c] = 0.0f;
#pragma unroll
    for (int x = 0; x < DD_CAP; ++x) { cvs[c][x] = -INFINITY; koff[c][x] = 0; }
  }
  // land the first step's scores before the loop: a load still pending at the loop header makes the
  // compiler wait with vmcnt(0) at the first use inside the loop, i.e. for the prefetch just issued
#pragma unroll
  for (int c = 0; c < W; ++c) asm volatile("" : "+v"(Sc[c]));
  float last = 0.0f, leftprev = 0.0f;
  bool ovf = false;
  const int nsteps = (int)L + (int)((L + W - 1) / W) - 1;  // the last lane that owns a column finishes row 0 here
  const int j0 = lane * W;
  float dknext[W <= 4 ? W : 1][DD_CAP];
#pragma unroll
  for (int c = 0; c < (W <= 4 ? W : 1); ++c)
#pragma unroll
    for (int x = 0; x < DD_CAP; ++x) dknext[c][x] = 0.0f;  // no candidates yet: every slot pairs with -inf
  // index of cell (i, i) in the packed triangle, kept by differences: tri(i-1) = tri(i) - (L - (i-1))
  uint32_t tbase = (uint32_t)(L - 1 + lane) * L - (uint32_t)(L - 1 + lane) * (uint32_t)(L - 2 + lane) / 2;
  uint32_t rslot = (uint32_t)(L - 1 + lane) % R;  // slot of this step's row, stepped down with the row
  for (int s = 0; s < nsteps; ++s) {
    const int i = (int)L - 1 - (s - lane);
    const bool rowv = i >= 0 && i < (int)L;
    if (s + 1 < nsteps) {
#pragma unroll
      for (int c = 0; c < W; ++c) nx[c] = S[((size_t)(s + 1) * W + c) * 64 + lane];
    }
    const float recv = wave_shr1(last);
    float diag = leftprev;
    float left = recv;
    const uint32_t ui = (uint32_t)i;
    DD_LDS char* rrow = ring + rslot * (L * 4);
    const int d0 = j0 - i;
    const uint32_t knew = ui ? (ui - 1) * 4 : 0u;
    // A candidate (k, j) has k <= j - 3, so the column k-1 it reads lies at least four columns to the left of
    // j: with up to four columns per lane it is never one of this lane's own columns of this step, and all
    // the dp[i][k-1] of the step can be fetched in one round trip before the first cell.
    // They are in fact fetched at the end of the previous step (dknext): what they read was written by lanes
    // to the left, which are at least one row ahead, so the values are there by then, and the wait for LDS is
    // spent on the step's preamble instead of in front of its first cell.
#pragma unroll
    for (int c = 0; c < W; ++c) {
      const int j = j0 + c, d = d0 + c;
      const bool pub = rowv && j < (int)L && d >= 0;  // cells whose value others read: dp[i][i] = dp[i][i+1] = 0 included
      const bool act = pub && d >= 2;
      const float below = P[c];
      float v = below;                             // nussinov.cpp:226-233
      uint32_t t = 1u;
      const bool m2 = v < left;
      v = m2 ? left : v; t = m2 ? 2u : t;
      const float sc = Sc[c];
      const float cand = diag + sc;                // :236
      const bool pos = sc > 0.0f;
      const bool m3 = pos && v < cand;
      v = m3 ? cand : v; t = m3 ? 3u : t;
      float dk[DD_CAP];
#pragma unroll
      for (int x = 0; x < DD_CAP; ++x) dk[x] = (W <= 4) ? dknext[c][x] : *(DD_LDS const float*)(rrow + koff[c][x]);  // all dp[i][k-1] in one round trip
#pragma unroll
      for (int x = 0; x < DD_CAP; ++x) {           // bifurcations, oldest candidate first (:245-255)
        const float cx = dk[x] + cvs[c][x];
        const bool m = v < cx;
        v = m ? cx : v; t = m ? (uint32_t)(4 + x) : t;
      }
      v = act ? v : 0.0f;
      t = act ? t : 0u;
      if (pub) {
        *(DD_LDS float*)(rrow + j * 4) = v;
        const uint32_t q = tbase + (uint32_t)d;
        if (TRG) trbg[q] = (uint8_t)t;
        else __hip_atomic_fetch_or(&trb[q >> 3], t << ((q & 7u) * 4), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
      if (act && pos) {                            // a new candidate for column j
        const uint32_t nc = n[c];
        if (nc < DD_CAP) lck[nc * L + j] = ui; else ovf = true;
#pragma unroll
        for (int x = 0; x < DD_CAP; ++x) {
          const bool here = nc == (uint32_t)x;
          cvs[c][x] = here ? cand : cvs[c][x];
          koff[c][x] = here ? knew : koff[c][x];
        }
        n[c] = nc + 1;
      }
      diag = below;
      P[c] = v;
      left = v;
    }
    leftprev = recv;
    last = left;
    tbase = tbase + ui - (L + 1);
#pragma unroll
    for (int c = 0; c < W; ++c) Sc[c] = nx[c];
    rslot = rslot ? rslot - 1 : R - 1;
    if (W <= 4) {  // the next step's dp[i-1][k-1], with this step's new candidates included
      DD_LDS const char* nrow = ring + rslot * (L * 4);
#pragma unroll
      for (int c = 0; c < (W <= 4 ? W : 1); ++c)
#pragma unroll
        for (int x = 0; x < DD_CAP; ++x) dknext[c][x] = *(DD_LDS const float*)(nrow + koff[c][x]);
    }
  }
  *ovf_out = __any(ovf);
  // dp[0][L-1]: what the lane that owns the last column holds after its last step
  float score = 0.0f;
#pragma unroll
  for (int c = 0; c < W; ++c)
    if ((uint32_t)c == (L - 1) % (uint32_t)W) score = P[c];
  return __shfl(score, (int)((L - 1) / W));
}

template <bool TRG>
__device__ __noinline__ float nuss_wave_fast_t(uint32_t W, uint32_t L, const float* S, uint32_t* trb, uint8_t* trbg, float* ring, uint32_t* lck, int lane, bool* ovf) {
  switch (W) {
    case 1: return nuss_wave_reg<1, TRG>(L, S, trb, trbg, ring, lck, lane, ovf);
    case 2: return nuss_wave_reg<2, TRG>(L, S, trb, trbg, ring, lck, lane, ovf);
    case 3: return nuss_wave_reg<3, TRG>(L, S, trb, trbg, ring, lck, lane, ovf);
    case 4: return nuss_wave_reg<4, TRG>(L, S, trb, trbg, ring, lck, lane, ovf);
    case 5: return nuss_wave_reg<5, TRG>(L, S, trb, trbg, ring, lck, lane, ovf);
    case 6: return nuss_wave_reg<6, TRG>(L, S, trb, trbg, ring, lck, lane, ovf);
    case 7: return nuss_wave_reg<7, TRG>(L, S, trb, trbg, ring, lck, lane, ovf);
    case 8: return nuss_wave_reg<8, TRG>(L, S, trb, trbg, ring, lck, lane, ovf);
    default: break;
  }
  if (TRG) {  // 9-16 columns per lane: alignments of 513-768 columns, whose codes never fit LDS
    switch (W) {
      case 9: return nuss_wave_reg<9, true>(L, S, trb, trbg, ring, lck, lane, ovf);
      case 10: return nuss_wave_reg<10, true>(L, S, trb, trbg, ring, lck, lane, ovf);
      case 11: return nuss_wave_reg<11, true>(L, S, trb, trbg, ring, lck, lane, ovf);
      case 12: return nuss_wave_reg<12, true>(L, S, trb, trbg, ring, lck, lane, ovf);
      case 13: return nuss_wave_reg<13, true>(L, S, trb, trbg, ring, lck, lane, ovf);
      case 14: return nuss_wave_reg<14, true>(L, S, trb, trbg, ring, lck, lane, ovf);
      case 15: return nuss_wave_reg<15, true>(L, S, trb, trbg, ring, lck, lane, ovf);
      case 16: return nuss_wave_reg<16, true>(L, S, trb, trbg, ring, lck, lane, ovf);
      default: break;
    }
  }
  *ovf = true;  // no register form for this width: the caller falls back to the span-ordered form
  return 0.0f;
}
// trb (LDS nibbles) when the fold was granted room for them, else the byte table trbg in HBM
__device__ __forceinline__ float nuss_wave_fast(uint32_t W, uint32_t L, const float* S, uint32_t* trb, uint8_t* trbg, float* ring, uint32_t* lck, int lane, bool* ovf) {
  return trb ? nuss_wave_fast_t<false>(W, L, S, trb, trbg, ring, lck, lane, ovf) : nuss_wave_fast_t<true>(W, L, S, trb, trbg, ring, lck, lane, ovf);
}

// ------------------------------------------------------------------------------------------
// Span-ordered single-wavefront folding DP ("span form"), for foldings whose whole triangle fits LDS.
// nuss_wave_reg gives every lane W columns and walks the rows: L + L/W steps of W cells, of which only the upper
// triangle is work -- 600 cell slots per lane at L = 150.  Here the lanes own ROWS (lane t: rows t, t+64, ...: NS
// slots) and time is the span d = j - i: L - 3 steps, and a slot is only visited while its row still has a cell
// of that span -- 257 slots per lane at L = 150.  What a cell needs:
//   dp[i+1][j]   (span d-1, row i+1)  the neighbour lane's value of the previous step: one DPP shift (lane 63 takes
//                                     lane 0's next slot through readfirstlane and a select)
//   dp[i][j-1]   (span d-1, row i)    this lane's own previous value
//   dp[i+1][j-1] (span d-2, row i+1)  the shifted value of the step before
//   dp[i][k-1] of the column's candidates: row i again -- a lane only ever reads dp rows it wrote itself, so the
//                                     table (row-major packed triangle in LDS) needs no ordering between lanes
//   the candidates of column j (values and row offsets, DD_CAP slots of 16 bytes each in LDS, empty = -inf): written
//   by the lanes that met them at earlier spans, fetched for the NEXT step's column j+1 at the end of a step, after
//   this step's insertions (the lane one row down may just have added to that very column).
// Codes, split rows (lck) and the overflow rule are those of nuss_wave_reg, so nuss_traceback_fast serves both.
// S is stored by span: S[d * Lp + i], Lp = L rounded up to 64 (fold_sidx).
template <int NS>
__device__ float nuss_wave_span(uint32_t L_, const float* S_, uint32_t* trb_, float* tri_, float* cval_, uint32_t* ckof_, uint32_t* lck_, int lane, bool* ovf_out) {
  // wave-uniform values in scalar registers: the compiler cannot see that what came out of the node descriptor is uniform,
  // and would mask every slot guard and address computation lane by lane
  const uint32_t L = (uint32_t)__builtin_amdgcn_readfirstlane((int)L_);
  DD_GLB const float* S = (DD_GLB const float*)S_;
  DD_LDS uint32_t* trb = (DD_LDS uint32_t*)trb_;  // one nibble per cell of the upper triangle, zeroed by the caller
  DD_LDS char* tri = (DD_LDS char*)tri_;          // dp, packed triangle by rows; spans 0..2 hold 0 (zeroed once per launch)
  DD_LDS char* cval = (DD_LDS char*)cval_;        // [L+1][DD_CAP] candidate values, -inf = empty (reset by the caller)
  DD_LDS char* ckof = (DD_LDS char*)ckof_;        // [L+1][DD_CAP] byte offset of dp[.][k-1] within a row
  DD_LDS uint32_t* lck = (DD_LDS uint32_t*)lck_;  // [DD_CAP][L] split row of candidate x of column j (traceback)
  static_assert(DD_CAP == 4, "the candidate slots are read as one 16-byte word");
  typedef float v4f __attribute__((ext_vector_type(4)));        // builtin vectors: loadable through address-space-qualified pointers
  typedef uint32_t v4u __attribute__((ext_vector_type(4)));
  const uint32_t Lp = (L + 63) & ~63u;
  float prev[NS], dg[NS];
  uint32_t qrow[NS], rb[NS];
  v4f cv[NS];
  v4u ck[NS];
  // A step is a few hundred nanoseconds, a load from L2/HBM longer: the scores are fetched PF steps ahead into a
  // rotating set of registers (the loop below is unrolled by PF so that the rotation is static).
  constexpr int PF = 4;
  float sq[PF][NS];
  auto fetch = [&](uint32_t dd, float (&dst)[NS]) {
    if (dd < L) {
#pragma unroll
      for (int r = 0; r < NS; ++r)
        if (64u * r < L - dd) dst[r] = S[(size_t)dd * Lp + (uint32_t)lane + 64u * r];
    }
  };
#pragma unroll
  for (int r = 0; r < NS; ++r) {
    const uint32_t i = (uint32_t)lane + 64u * r;
    const uint32_t ic = i < L ? i : 0u;
    qrow[r] = (uint32_t)tri_index(L, ic, ic);
    rb[r] = (qrow[r] - ic) * 4u;
    prev[r] = 0.0f; dg[r] = 0.0f;
#pragma unroll
    for (int k = 0; k < PF; ++k) sq[k][r] = 0.0f;
    const uint32_t jn = i + 3 < L ? i + 3 : L;  // entry L: the always-empty list
    cv[r] = *(DD_LDS const v4f*)(cval + jn * 16u);
    ck[r] = *(DD_LDS const v4u*)(ckof + jn * 16u);
  }
#pragma unroll
  for (int k = 0; k < PF; ++k) fetch(3u + k, sq[k]);
  bool ovf = false;
  // One span.  NA = row slots that still have a cell of this span (compile time: the caller branches on the span once
  // per step, a scalar branch, and the slots need no guards of their own).
  auto step = [&](auto na_tag, uint32_t d, const float (&sc)[NS]) {
    constexpr int NA = decltype(na_tag)::value;
    const uint32_t ncell = L - d;  // rows 0 .. ncell-1 have a cell of this span
    float dk[NA][DD_CAP], below[NA];
#pragma unroll
    for (int r = 0; r < NA; ++r) {
      DD_LDS const char* row = tri + rb[r];
      dk[r][0] = *(DD_LDS const float*)(row + ck[r].x);
      dk[r][1] = *(DD_LDS const float*)(row + ck[r].y);
      dk[r][2] = *(DD_LDS const float*)(row + ck[r].z);
      dk[r][3] = *(DD_LDS const float*)(row + ck[r].w);
    }
#pragma unroll
    for (int r = 0; r < NA; ++r) {
      int b = __builtin_amdgcn_update_dpp(0, __float_as_int(prev[r]), 0x130, 0xf, 0xf, false);  // wave_shl:1: lane l takes lane l+1
      if (r + 1 < NS) {  // read outside the select: inside it the read would run with lane 63 alone and return lane 63's value
        const int first = __builtin_amdgcn_readfirstlane(__float_as_int(prev[r + 1 < NS ? r + 1 : r]));
        b = lane == 63 ? first : b;
      }
      below[r] = __int_as_float(b);
    }
#pragma unroll
    for (int r = 0; r < NA; ++r) {
      const uint32_t i = (uint32_t)lane + 64u * r;
      const bool valid = i < ncell;
      const uint32_t j = i + d;
      float v = below[r];                       // nussinov.cpp:226-233
      uint32_t t = 1u;
      const bool m2 = v < prev[r];
      v = m2 ? prev[r] : v; t = m2 ? 2u : t;
      const float s = sc[r];
      const float cand = dg[r] + s;             // :236
      const bool pos = s > 0.0f;
      const bool m3 = pos && v < cand;
      v = m3 ? cand : v; t = m3 ? 3u : t;
      const float cvx[DD_CAP] = {cv[r].x, cv[r].y, cv[r].z, cv[r].w};
#pragma unroll
      for (int x = 0; x < DD_CAP; ++x) {        // bifurcations, oldest candidate first (:245-255)
        const float cx = dk[r][x] + cvx[x];
        const bool m = v < cx;
        v = m ? cx : v; t = m ? (uint32_t)(4 + x) : t;
      }
      if (valid) {
        const uint32_t q = qrow[r] + d;
        *(DD_LDS float*)(tri + q * 4u) = v;
        __hip_atomic_fetch_or(&trb[q >> 3], t << ((q & 7u) * 4), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
      if (valid && pos) {                       // a new candidate for column j (slots fill in order: count the occupied ones)
        const uint32_t nc = (cvx[0] > -INFINITY ? 1u : 0u) + (cvx[1] > -INFINITY ? 1u : 0u) + (cvx[2] > -INFINITY ? 1u : 0u) + (cvx[3] > -INFINITY ? 1u : 0u);
        if (nc < DD_CAP) {
          *(DD_LDS float*)(cval + j * 16u + nc * 4u) = cand;
          *(DD_LDS uint32_t*)(ckof + j * 16u + nc * 4u) = i ? (i - 1) * 4u : 0u;
          lck[nc * L + j] = i;
        } else ovf = true;
      }
      dg[r] = below[r];
      prev[r] = v;
    }
    wave_lds_fence();
#pragma unroll
    for (int r = 0; r < NA; ++r) {  // the lists of the next step's columns (a slot on its last span reads the empty list)
      const uint32_t jn = (uint32_t)lane + 64u * r + d + 1;
      const uint32_t jc = jn < L ? jn : L;
      cv[r] = *(DD_LDS const v4f*)(cval + jc * 16u);
      ck[r] = *(DD_LDS const v4u*)(ckof + jc * 16u);
    }
  };
  for (uint32_t d0 = 3; d0 < L; d0 += PF) {
#pragma unroll
    for (int k = 0; k < PF; ++k) {
      const uint32_t d = d0 + k;
      if (d < L) {
        float cur[NS];
#pragma unroll
        for (int r = 0; r < NS; ++r) cur[r] = sq[k][r];
        fetch(d + PF, sq[k]);
        const uint32_t ncell = L - d;
        if (NS >= 4 && ncell > 192) step(std::integral_constant<int, (NS >= 4 ? 4 : NS)>(), d, cur);
        else if (NS >= 3 && ncell > 128) step(std::integral_constant<int, (NS >= 3 ? 3 : NS)>(), d, cur);
        else if (NS >= 2 && ncell > 64) step(std::integral_constant<int, (NS >= 2 ? 2 : NS)>(), d, cur);
        else step(std::integral_constant<int, 1>(), d, cur);
      }
    }
  }
  *ovf_out = __any(ovf);
  return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(prev[0])));  // dp[0][L-1]: row 0, last span
}

// The span form on one wavefront PER ROW SLOT (round 3), for a folding that has a workgroup of its own (split mode): wave r
// owns rows 64 r .. 64 r + 63, all waves of the workgroup walk the spans together, one workgroup barrier per span.  The
// single wavefront above visits its row slots one after the other -- sum over the slots of the spans they live, 297 slot
// steps at L = 163 --; here a span costs one slot step plus the barrier, L - 3 of them.  What crosses the wavefronts goes
// through LDS, which holds it anyway: lane 63 reads dp[i+1][j] of the slot above from the triangle (wave r + 1 wrote it in
// the previous span), and the candidate lists of the next span's columns are fetched after the barrier, when every wave's
// insertions of this span are in.  (Round 2's attempt kept the waves in step with LDS mailboxes they polled: the polling
// slowed the LDS for the waves at work.  s_barrier costs nothing while waiting.)  Same cells, same order per cell, same
// codes, candidate lists and overflow rule as nuss_wave_span: bit-identical, and nuss_traceback_fast serves both.
// Every wave of the workgroup must call it (waves without a slot only keep the barriers).  Returns dp[0][L-1] in wave 0.
__device__ __noinline__ float nuss_span_mw(uint32_t L_, const float* S_, uint32_t* trb_, float* tri_, float* cval_, uint32_t* ckof_, uint32_t* lck_, int wave, int lane,
                                           bool* ovf_out) {
  const uint32_t L = (uint32_t)__builtin_amdgcn_readfirstlane((int)L_);
  const uint32_t r = (uint32_t)__builtin_amdgcn_readfirstlane(wave);
  DD_GLB const float* S = (DD_GLB const float*)S_;
  DD_LDS uint32_t* trb = (DD_LDS uint32_t*)trb_;
  DD_LDS char* tri = (DD_LDS char*)tri_;
  DD_LDS char* cval = (DD_LDS char*)cval_;
  DD_LDS char* ckof = (DD_LDS char*)ckof_;
  DD_LDS uint32_t* lck = (DD_LDS uint32_t*)lck_;
  typedef float v4f __attribute__((ext_vector_type(4)));
  typedef uint32_t v4u __attribute__((ext_vector_type(4)));
  const uint32_t Lp = (L + 63) & ~63u;
  const uint32_t i = (uint32_t)lane + 64u * r;       // this lane's row
  const bool slot = 64u * r < L;                     // (scalar) this wave has rows at all
  const uint32_t ic = i < L ? i : 0u;
  const uint32_t qrow = (uint32_t)tri_index(L, ic, ic);
  const uint32_t rb = (qrow - ic) * 4u;
  // dp[i+1][j] of lane 63's neighbour row 64 (r + 1): cell (i + 1, i + 1 + (d - 1)) of the triangle
  const uint32_t inext = i + 1 < L ? i + 1 : 0u;
  const uint32_t qnext = (uint32_t)tri_index(L, inext, inext);
  float prev = 0.0f, dg = 0.0f;
  constexpr int PF = 4;
  float sq[PF];
#pragma unroll
  for (int k = 0; k < PF; ++k) {
    sq[k] = 0.0f;
    if (slot && 3u + k < L && i < L - (3u + k)) sq[k] = S[(size_t)(3u + k) * Lp + i];
  }
  bool ovf = false;
  v4f cv;
  v4u ck;
  {
    const uint32_t jn = i + 3 < L ? i + 3 : L;  // entry L: the always-empty list
    cv = *(DD_LDS const v4f*)(cval + jn * 16u);
    ck = *(DD_LDS const v4u*)(ckof + jn * 16u);
  }
  for (uint32_t d0 = 3; d0 < L; d0 += PF) {
#pragma unroll
    for (int k = 0; k < PF; ++k) {
      const uint32_t d = d0 + k;
      if (d >= L) break;
      const uint32_t ncell = L - d;               // rows 0 .. ncell - 1 have a cell of this span
      if (slot && 64u * r < ncell) {              // (scalar) this slot still lives
        const float s = sq[k];
        if (d + PF < L) sq[k] = (i < L - (d + PF)) ? S[(size_t)(d + PF) * Lp + i] : 0.0f;
        DD_LDS const char* row = tri + rb;
        const float dk0 = *(DD_LDS const float*)(row + ck.x), dk1 = *(DD_LDS const float*)(row + ck.y);
        const float dk2 = *(DD_LDS const float*)(row + ck.z), dk3 = *(DD_LDS const float*)(row + ck.w);
        const float across = *(DD_LDS const float*)(tri + (qnext + (d - 1)) * 4u);  // lane 63's neighbour lives in the next wave
        int b = __builtin_amdgcn_update_dpp(0, __float_as_int(prev), 0x130, 0xf, 0xf, false);  // wave_shl:1: lane l takes lane l+1
        const float below = lane == 63 ? across : __int_as_float(b);
        const bool valid = i < ncell;
        const uint32_t j = i + d;
        float v = below;                          // nussinov.cpp:226-233
        uint32_t t = 1u;
        const bool m2 = v < prev;
        v = m2 ? prev : v; t = m2 ? 2u : t;
        const float cand = dg + s;                // :236
        const bool pos = s > 0.0f;
        const bool m3 = pos && v < cand;
        v = m3 ? cand : v; t = m3 ? 3u : t;
        const float cvx[DD_CAP] = {cv.x, cv.y, cv.z, cv.w};
        const float dk[DD_CAP] = {dk0, dk1, dk2, dk3};
#pragma unroll
        for (int x = 0; x < DD_CAP; ++x) {        // bifurcations, oldest candidate first (:245-255)
          const float cx = dk[x] + cvx[x];
          const bool m = v < cx;
          v = m ? cx : v; t = m ? (uint32_t)(4 + x) : t;
        }
        if (valid) {
          const uint32_t q = qrow + d;
          *(DD_LDS float*)(tri + q * 4u) = v;
          __hip_atomic_fetch_or(&trb[q >> 3], t << ((q & 7u) * 4), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        if (valid && pos) {                       // a new candidate for column j (slots fill in order: count the occupied ones)
          const uint32_t nc = (cvx[0] > -INFINITY ? 1u : 0u) + (cvx[1] > -INFINITY ? 1u : 0u) + (cvx[2] > -INFINITY ? 1u : 0u) + (cvx[3] > -INFINITY ? 1u : 0u);
          if (nc < DD_CAP) {
            *(DD_LDS float*)(cval + j * 16u + nc * 4u) = cand;
            *(DD_LDS uint32_t*)(ckof + j * 16u + nc * 4u) = i ? (i - 1) * 4u : 0u;
            lck[nc * L + j] = i;
          } else ovf = true;
        }
        dg = below;
        prev = v;
      }
      __syncthreads();                            // this span's values and insertions are in
      if (slot) {                                 // the lists of the next span's columns (a row on its last span reads the empty list)
        const uint32_t jn = i + d + 1;
        const uint32_t jc = jn < L ? jn : L;
        cv = *(DD_LDS const v4f*)(cval + jc * 16u);
        ck = *(DD_LDS const v4u*)(ckof + jc * 16u);
      }
    }
  }
  *ovf_out = __any(ovf);
  return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(prev)));  // wave 0: dp[0][L-1] (row 0, last span)
}

__device__ __noinline__ float nuss_wave_span_t(uint32_t L, const float* S, uint32_t* trb, float* tri, float* cval, uint32_t* ckof, uint32_t* lck, int lane, bool* ovf) {
  switch ((L + 63) / 64) {
    case 1: return nuss_wave_span<1>(L, S, trb, tri, cval, ckof, lck, lane, ovf);
    case 2: return nuss_wave_span<2>(L, S, trb, tri, cval, ckof, lck, lane, ovf);
    case 3: return nuss_wave_span<3>(L, S, trb, tri, cval, ckof, lck, lane, ovf);
    case 4: return nuss_wave_span<4>(L, S, trb, tri, cval, ckof, lck, lane, ovf);
    default: break;
  }
  *ovf = true;  // the planner (capi_dd.cpp) grants the span form up to DD_SPAN_LMAX columns only
  return 0.0f;
}

// Traceback of nuss_wave_reg's codes by the whole wavefront.  The walk itself is sequential, but it consists
// of runs: stretches of code 1 (i+1), of code 2 (j-1) and stacks of code 3 (i+1, j-1).  The lanes read the
// next 64 cells along the current direction at once and a ballot finds where the run ends, so a run costs
// two LDS round trips instead of two per cell.  Bifurcations park their left half on an LDS stack.
__device__ void nuss_traceback_fast(uint32_t L, uint32_t* trb_, const uint8_t* trbg_, uint32_t* lck_, uint32_t* ss_, uint32_t* stack_, int lane) {
  DD_LDS const uint32_t* trb = (DD_LDS const uint32_t*)trb_;
  const bool in_lds = trb_ != nullptr;
  if (!in_lds) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");  // the DP's code stores have reached L2
  DD_LDS const uint32_t* lck = (DD_LDS const uint32_t*)lck_;
  DD_LDS uint32_t* stack = (DD_LDS uint32_t*)stack_;
  DD_GLB uint32_t* ss = (DD_GLB uint32_t*)ss_;
  auto code = [&](int i, int j) -> uint32_t {  // traceback code of cell (i, j), 0 outside the triangle j > i
    if (!(i >= 0 && j > i && j < (int)L)) return 0u;
    const uint32_t q = (uint32_t)tri_index(L, (uint32_t)i, (uint32_t)j);
    // the byte table was written by other lanes: read it at L2 (agent scope), past this CU's vector cache
    return in_lds ? (trb[q >> 3] >> ((q & 7u) * 4)) & 15u
                  : (uint32_t)__hip_atomic_load((const uint8_t*)trbg_ + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  uint32_t sp = 0;
  int i = 0, j = (int)L - 1;
  uint32_t guard = 4 * L + 8;
  while (guard--) {
    const uint32_t t = (uint32_t)__builtin_amdgcn_readfirstlane((int)code(i, j));
    if (t == 0) {
      if (!sp) break;
      --sp;
      const uint32_t e = (uint32_t)__builtin_amdgcn_readfirstlane((int)stack[sp]);
      i = (int)(e >> 16); j = (int)(e & 0xFFFFu);
      continue;
    }
    if (t == 1) {         // run of i+1: land on the first cell below whose code is not 1
      const unsigned long long m = __ballot(code(i + 1 + lane, j) != 1u);
      i += 1 + (m ? (int)__ffsll((long long)m) - 1 : 64);
    } else if (t == 2) {  // run of j-1
      const unsigned long long m = __ballot(code(i, j - 1 - lane) != 2u);
      j -= 1 + (m ? (int)__ffsll((long long)m) - 1 : 64);
    } else if (t == 3) {  // stack: (i, j) pairs, and so does every further cell (i+1+l, j-1-l) whose code is 3
      const unsigned long long m = __ballot(code(i + 1 + lane, j - 1 - lane) != 3u);
      const int r = m ? (int)__ffsll((long long)m) - 1 : 64;
      if (lane == 0) ss[i] = (uint32_t)j;
      if (lane < r) ss[i + 1 + lane] = (uint32_t)(j - 1 - lane);
      i += 1 + r; j -= 1 + r;
    } else {
      const int k = (int)__builtin_amdgcn_readfirstlane((int)lck[(t - 4) * L + (uint32_t)j]);
      if (lane == 0) {
        ss[k] = (uint32_t)j;
        if (k - 1 > i) stack[sp] = ((uint32_t)i << 16) | (uint32_t)(k - 1);
      }
      if (k - 1 > i) ++sp;
      wave_lds_fence();
      i = k + 1; --j;
    }
  }
}

// The walk over the two-bit table by the whole wavefront (cf. nuss_traceback_fast): it consists of runs -- stretches
// of M (i-1, k-1), of X (i-1) and of Y (k-1).  The lanes read the next 64 cells along the current direction at once, a
// ballot finds where the run ends, and the run's entries of `al` are written together: two LDS round trips per run
// instead of one per cell and a store each.  Row 0 / column 0 are implicit (Y / X); (0,0) ends the walk.
// SLOTS: the table of nw_wave_reg<W, 2> in global memory -- one 64-bit slot per (panel, row, lane), the lane's W cells two bits each.
template <bool SLOTS>
__device__ bool nw_traceback_wave(uint32_t L1, uint32_t L2, const uint32_t* tr_, uint32_t* al_, int lane, uint32_t W = 1) {
  DD_LDS const uint32_t* tr = (DD_LDS const uint32_t*)tr_;
  DD_GLB const unsigned long long* slots = (DD_GLB const unsigned long long*)tr_;
  DD_GLB uint32_t* al = (DD_GLB uint32_t*)al_;
  const uint32_t RW = dd_nwtab_row_words(L2);  // words per row of the table
  auto code = [&](int i, int k) -> uint32_t {  // 1 M, 2 X, 3 Y; 0 at (0,0), outside the grid and where the DP left no mark
    if (i < 0 || k < 0 || (i == 0 && k == 0)) return 0u;
    if (i == 0) return 3u;
    if (k == 0) return 2u;
    if constexpr (SLOTS) {
      const uint32_t panel = (uint32_t)k / (64u * W), kp = (uint32_t)k - panel * 64u * W;
      const uint32_t owner = kp / W, c = kp - owner * W;
      return (uint32_t)(slots[((size_t)panel * (L1 + 1) + (uint32_t)i) * 64 + owner] >> (2 * c)) & 3u;
    }
    return (tr[(uint32_t)i * RW + ((uint32_t)k >> 4)] >> (((uint32_t)k & 15u) * 2)) & 3u;
  };
  int i = (int)L1, k = (int)L2;
  uint32_t guard = L1 + L2 + 2;
  // the code of the current cell; after a run it is what the lane that found the run's end has just read
  uint32_t t = (uint32_t)__builtin_amdgcn_readfirstlane((int)code(i, k));
  while ((i > 0 || k > 0) && guard--) {
    uint32_t probe;
    unsigned long long m;
    if (t == 1u) {         // (i,k), (i-1,k-1), ...: the run ends at the first cell whose code is not M
      probe = code(i - 1 - lane, k - 1 - lane);
      m = __ballot(probe != 1u);
      const int r = m ? (int)__ffsll((long long)m) - 1 : 63;  // cells of the run after the first (at most 64 cells a turn)
      if (lane <= r) al[i - 1 - lane] = (uint32_t)(k - 1 - lane);
      i -= 1 + r; k -= 1 + r;
      t = m ? (uint32_t)__builtin_amdgcn_readlane((int)probe, r) : (uint32_t)__builtin_amdgcn_readfirstlane((int)code(i, k));
    } else if (t == 2u) {  // X run: rows i, i-1, ... of column k
      probe = code(i - 1 - lane, k);
      m = __ballot(probe != 2u);
      const int r = m ? (int)__ffsll((long long)m) - 1 : 63;
      if (lane <= r) al[i - 1 - lane] = DD_NONE;
      i -= 1 + r;
      t = m ? (uint32_t)__builtin_amdgcn_readlane((int)probe, r) : (uint32_t)__builtin_amdgcn_readfirstlane((int)code(i, k));
    } else if (t == 3u) {  // Y run
      probe = code(i, k - 1 - lane);
      m = __ballot(probe != 3u);
      const int r = m ? (int)__ffsll((long long)m) - 1 : 63;
      k -= 1 + r;
      t = m ? (uint32_t)__builtin_amdgcn_readlane((int)probe, r) : (uint32_t)__builtin_amdgcn_readfirstlane((int)code(i, k));
    } else return false;
  }
  return i == 0 && k == 0;
}

// MODE 1: codes two bits per cell in an LDS table whose rows are whole words (OR-ed in; zeroed by the caller); one panel.
// MODE 2: codes in global memory, one 64-bit slot per (panel, row, lane) holding the lane's W cells (plain stores, nothing to
// zero).  Second alignments of more than 64 W columns run as PANELS of 64 W columns, one after the other over all the
// rows: the last column of a panel goes to `edge` (two arrays of L1 + 2 floats, used in turn), where lane 0 of the next panel
// finds its left neighbours -- dp[i][k-1] for the row it is on, and through it dp[i-1][k-1] one step later.  The inputs are
// stored panel by panel in sweep order (nw_idx).
template <int W, int MODE>
__device__ float nw_wave_reg(uint32_t L1_, uint32_t L2_, const float* ps_, const float* qs_, float th_, const uint32_t* env_, uint8_t* tr_, float* edge_, int lane) {
  // wave-uniform values in scalar registers (see nuss_wave_span)
  const uint32_t L1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)L1_), L2 = (uint32_t)__builtin_amdgcn_readfirstlane((int)L2_);
  const float th = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(th_)));
  DD_GLB const uint32_t* env = (DD_GLB const uint32_t*)env_;
  DD_LDS uint32_t* tr_l = (DD_LDS uint32_t*)tr_;  // MODE 1: two bits per cell (1 M, 2 X, 3 Y), zeroed by the caller
  DD_GLB float* edge = (DD_GLB float*)edge_;
  typedef uint32_t v2u __attribute__((ext_vector_type(2)));
  const uint32_t npanels = MODE == 1 ? 1u : (L2 + 64u * W) / (64u * W);  // columns 0 .. L2
  const size_t panel_words = ((size_t)L1 + 63) * W * 64;                  // inputs of one panel
  const bool lane0 = lane == 0;
  // Input register sets.  AHEAD = how many steps ahead a step's inputs are fetched: the narrow LDS-table forms fetch three
  // steps ahead into four sets that take turns (a step of ~0.2 us is shorter than the trip to L2, and a wait at the end of
  // every step for the loads issued at its start made the step exactly that trip long); the wide forms keep one step.
  constexpr int AHEAD = (MODE == 1 && W <= 8) ? 3 : 1;
  float P[W], B[AHEAD + 1][2][W];
  v2u E[AHEAD + 1];  // the envelope {lo, hi} of the row each set's step is on: fetched as far ahead as the inputs
  float score = 0.0f;
  for (uint32_t panel = 0; panel < npanels; ++panel) {
    const int kbase = (int)(panel * 64u * W);
    const bool first = panel == 0;
    DD_GLB const float* ps = (DD_GLB const float*)ps_ + panel * panel_words;
    DD_GLB const float* qs = (DD_GLB const float*)qs_ + panel * panel_words;
    DD_GLB unsigned long long* tr_g = (DD_GLB unsigned long long*)tr_ + (size_t)panel * (L1 + 1) * 64;
    DD_GLB const float* ein = edge + (panel & 1u ? 0 : (L1 + 2));   // written by the panel before
    DD_GLB float* eout = edge + (panel & 1u ? (L1 + 2) : 0);
#pragma unroll
    for (int c = 0; c < W; ++c) P[c] = 0.0f;
    float last = 0.0f, leftprev = 0.0f;
    const uint32_t lanes_on = (L2 - (uint32_t)kbase) / W + 1 < 64u ? (L2 - (uint32_t)kbase) / W + 1 : 64u;  // lanes beyond column L2 have nothing to do
    const int nsteps = (int)L1 + (int)lanes_on - 1;
    float e_cur = 0.0f;  // lane 0: dp[row of this step][kbase - 1]
    if (!first) e_cur = ein[1];
#pragma unroll
    for (int a = 0; a < AHEAD; ++a)   // the inputs of steps 0 .. AHEAD-1
#pragma unroll
      for (int c = 0; c < W; ++c) {
        B[a][0][c] = a < nsteps ? ps[((size_t)a * W + c) * 64 + lane] : 0.0f;
        B[a][1][c] = a < nsteps ? qs[((size_t)a * W + c) * 64 + lane] : 0.0f;
      }
#pragma unroll
    for (int a = 0; a < AHEAD; ++a) E[a] = ((DD_GLB const v2u*)env)[1 - lane + a + 64];  // rows of steps 0 .. AHEAD-1 (the padding makes every index valid)
    asm volatile("" : "+v"(e_cur));
    // Straight-line steps: every cell is computed and then replaced by what its place in the grid says (outside the
    // envelope: lowest(); column 0: 0; rows outside the grid: unchanged), and the codes of the lane's W cells -- consecutive
    // cells of one row -- go out together.  What a step needs besides its cells is kept small: env is the padded envelope
    // (dd_node::env4: {max(first,1), second} of row r at r + 64, empty ranges around the grid: one 8-byte load, no
    // clamping); the LDS table's rows are whole words, so the lane's shift and word offset never change and its word
    // pointer just moves down a row.
    const int k0 = kbase + lane * W;
    const uint32_t RW = dd_nwtab_row_words(L2);
    const uint32_t sh = ((uint32_t)k0 & 15u) * 2u;
    int i = 1 - lane;                                                      // row of step 0
    DD_GLB const v2u* envp = (DD_GLB const v2u*)env + (i + AHEAD + 64);    // envelope of the row of step AHEAD
    DD_LDS uint32_t* word = tr_l + (int)RW * i + (k0 >> 4);                // MODE 1: the lane's cells in row i (never touched while i < 1)
    DD_GLB unsigned long long* slot = tr_g + (ptrdiff_t)i * 64 + lane;     // MODE 2: the lane's slot of row i
    DD_GLB const float* pnext = ps + (size_t)AHEAD * W * 64 + lane;        // inputs of step AHEAD
    DD_GLB const float* qnext = qs + (size_t)AHEAD * W * 64 + lane;
    // One step: cp / cq hold its inputs, xp / xq receive those of step s + AHEAD.
    auto step = [&](int s, float (&cp)[W], float (&cq)[W], const v2u& ce, float (&xp)[W], float (&xq)[W], v2u& xe) __attribute__((always_inline)) {
      const int lo = (int)ce.x, hi = (int)ce.y;
      if (s + AHEAD < nsteps) xe = *envp;
      float e_next = 0.0f;
      if (MODE == 2 && !first) e_next = ein[(uint32_t)(s + 2) <= L1 ? s + 2 : (int)L1];  // lane 0's left neighbour of the next step
      __builtin_amdgcn_sched_barrier(0);
      if (s + AHEAD < nsteps) {
#pragma unroll
        for (int c = 0; c < W; ++c) { xp[c] = pnext[c * 64]; xq[c] = qnext[c * 64]; }
      }
      __builtin_amdgcn_sched_barrier(0);
      const bool rowv = i >= 1 && i <= (int)L1;
      const bool some = hi >= lo;                                   // an empty range (hi < lo: rows outside the grid) admits no k
      const int lo_e = some ? lo : 0x7fffffff;
      const uint32_t span = some ? (uint32_t)(hi - lo) : 0u;
      float recv = wave_shr1(last);  // lane 0 of the first panel owns column 0, which takes nothing from its left
      if (MODE == 2 && !first) recv = lane0 ? e_cur : recv;
      float diag = leftprev;
      float left = recv;
      float v = 0.0f;
      unsigned long long codes = 0;
#pragma unroll
      for (int c = 0; c < W; ++c) {
        const int k = k0 + c;
        const float up = P[c];
        float cand = diag + cp[c] - th;  // needleman_wunsch.cpp:281-283
        cand = cand + cq[c];
        const bool m1 = cand < up;
        const float v1 = m1 ? up : cand;
        const bool m2 = v1 < left;
        const float v2 = m2 ? left : v1;
        const uint32_t t = m2 ? 3u : (m1 ? 2u : 1u);
        const bool inside = (uint32_t)(k - lo_e) <= span;  // lo <= k <= hi in one compare; empty outside the rows of the grid
        v = inside ? v2 : -FLT_MAX;
        if (c == 0) v = (lane0 && first) ? 0.0f : v;  // column 0
        v = rowv ? v : up;                       // rows outside the grid: the cell keeps what it held (row 0 / the last row)
        codes |= (unsigned long long)(inside ? t : 0u) << (2 * c);  // (cells beyond column L2 must leave the table's next row alone)
        diag = up;
        P[c] = v;
        left = v;
      }
      if constexpr (MODE == 1) {
        if (codes) {
          const unsigned long long w = codes << sh;
          __hip_atomic_fetch_or(word, (uint32_t)w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          if ((uint32_t)(w >> 32)) __hip_atomic_fetch_or(word + 1, (uint32_t)(w >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
      } else {
        if (rowv) *slot = codes;  // every slot of the grid is written in every pass: no stale codes
        if (rowv && lane == 63 && panel + 1 < npanels) eout[i] = v;  // the panel's last column, for the next panel
      }
      leftprev = recv;
      last = v;
      e_cur = e_next;
      ++i; ++envp; word += RW; slot += 64; pnext += W * 64; qnext += W * 64;
    };
    // the sets take turns: step s reads set s mod (AHEAD + 1) and fills the set the step before it has just used up
    int s = 0;
    if constexpr (AHEAD == 3) {
      for (; s + 3 < nsteps; s += 4) {
        step(s, B[0][0], B[0][1], E[0], B[3][0], B[3][1], E[3]);
        step(s + 1, B[1][0], B[1][1], E[1], B[0][0], B[0][1], E[0]);
        step(s + 2, B[2][0], B[2][1], E[2], B[1][0], B[1][1], E[1]);
        step(s + 3, B[3][0], B[3][1], E[3], B[2][0], B[2][1], E[2]);
      }
      if (s < nsteps) step(s, B[0][0], B[0][1], E[0], B[3][0], B[3][1], E[3]);
      if (s + 1 < nsteps) step(s + 1, B[1][0], B[1][1], E[1], B[0][0], B[0][1], E[0]);
      if (s + 2 < nsteps) step(s + 2, B[2][0], B[2][1], E[2], B[1][0], B[1][1], E[1]);
    } else {
      for (; s + 1 < nsteps; s += 2) {
        step(s, B[0][0], B[0][1], E[0], B[1][0], B[1][1], E[1]);
        step(s + 1, B[1][0], B[1][1], E[1], B[0][0], B[0][1], E[0]);
      }
      if (s < nsteps) step(s, B[0][0], B[0][1], E[0], B[1][0], B[1][1], E[1]);
    }
    if (panel + 1 < npanels) __threadfence_block();  // the edge column is in place before lane 0 of the next panel reads it
    else {
      // dp[L1][L2]: what the lane that owns column L2 holds there after its last row (rows beyond L1 leave P untouched)
      const uint32_t kk = L2 - (uint32_t)kbase;
#pragma unroll
      for (int c = 0; c < W; ++c)
        if ((uint32_t)c == kk % (uint32_t)W) score = P[c];
      score = __shfl(score, (int)(kk / W));
    }
  }
  return score;
}

template <int MODE>
__device__ __noinline__ float nw_wave_fast(uint32_t W, uint32_t L1, uint32_t L2, const float* ps, const float* qs, float th, const uint32_t* env, uint8_t* tr, float* edge, int lane) {
  switch (W) {
    case 1: return nw_wave_reg<1, MODE>(L1, L2, ps, qs, th, env, tr, edge, lane);
    case 2: return nw_wave_reg<2, MODE>(L1, L2, ps, qs, th, env, tr, edge, lane);
    case 3: return nw_wave_reg<3, MODE>(L1, L2, ps, qs, th, env, tr, edge, lane);
    case 4: return nw_wave_reg<4, MODE>(L1, L2, ps, qs, th, env, tr, edge, lane);
    case 5: return nw_wave_reg<5, MODE>(L1, L2, ps, qs, th, env, tr, edge, lane);
    case 6: return nw_wave_reg<6, MODE>(L1, L2, ps, qs, th, env, tr, edge, lane);
    case 7: return nw_wave_reg<7, MODE>(L1, L2, ps, qs, th, env, tr, edge, lane);
    case 8: return nw_wave_reg<8, MODE>(L1, L2, ps, qs, th, env, tr, edge, lane);
    case 9: return nw_wave_reg<9, MODE>(L1, L2, ps, qs, th, env, tr, edge, lane);
    case 10: return nw_wave_reg<10, MODE>(L1, L2, ps, qs, th, env, tr, edge, lane);
    case 11: return nw_wave_reg<11, MODE>(L1, L2, ps, qs, th, env, tr, edge, lane);
    case 12: return nw_wave_reg<12, MODE>(L1, L2, ps, qs, th, env, tr, edge, lane);
    case 13: return nw_wave_reg<13, MODE>(L1, L2, ps, qs, th, env, tr, edge, lane);
    case 14: return nw_wave_reg<14, MODE>(L1, L2, ps, qs, th, env, tr, edge, lane);
    case 15: return nw_wave_reg<15, MODE>(L1, L2, ps, qs, th, env, tr, edge, lane);
    case 16: return nw_wave_reg<16, MODE>(L1, L2, ps, qs, th, env, tr, edge, lane);  // DD_WNW
  }
  if constexpr (MODE == 2) {  // beyond DD_WNW: multiples of four up to DD_WNWG (dd_nw_cols)
    switch (W) {
      case 20: return nw_wave_reg<20, 2>(L1, L2, ps, qs, th, env, tr, edge, lane);
      case 24: return nw_wave_reg<24, 2>(L1, L2, ps, qs, th, env, tr, edge, lane);
      case 28: return nw_wave_reg<28, 2>(L1, L2, ps, qs, th, env, tr, edge, lane);
      case 32: return nw_wave_reg<32, 2>(L1, L2, ps, qs, th, env, tr, edge, lane);
    }
  }
  return 0.0f;  // unreachable: the caller asks for widths dd_nw_cols produces
}

// Sweep-order ("skewed") copies of the DP inputs: the value lane t needs at step s for its column c sits
// at ((s*W + c)*64 + t), so each step is one coalesced load per column.
__device__ __forceinline__ size_t nuss_skew(uint32_t L, uint32_t W, uint32_t i, uint32_t j) {
  const uint32_t lane = j / W, c = j - lane * W, step = L - 1 - i + lane;
  return ((size_t)step * W + c) * 64 + lane;
}
__device__ __forceinline__ size_t nw_skew(uint32_t W, uint32_t i, uint32_t k) {  // i in 1..L1, k in 0..64W-1 (within a panel)
  const uint32_t lane = k / W, c = k - lane * W, step = i - 1 + lane;
  return ((size_t)step * W + c) * 64 + lane;
}
// where input (i, k) of the alignment DP lives: panel k / 64W, then sweep order within the panel
__device__ __forceinline__ size_t nw_idx(uint32_t L1, uint32_t W, uint32_t i, uint32_t k) {  // i in 1..L1, k in 0..L2
  const uint32_t panel = k / (64u * W);
  return (size_t)panel * ((size_t)L1 + 63) * W * 64 + nw_skew(W, i, k - panel * 64u * W);
}
// where the score of cell (i, j), j >= i, lives: by span for the span form (nuss_wave_span), else in sweep order
__device__ __forceinline__ size_t fold_sidx(bool span, uint32_t L, uint32_t W, uint32_t i, uint32_t j) {
  return span ? (size_t)(j - i) * ((L + 63) & ~63u) + i : nuss_skew(L, W, i, j);
}
// all threads: S = w*(p-th)-q (nussinov.cpp:236); the association is the reference's
__device__ void dd_fill_scores(bool span, uint32_t L, const float* __restrict__ p, const float* __restrict__ q, float w, float th, float* S) {
  const uint32_t W = dd_fold_cols(L);
  for (size_t c = threadIdx.x; c < (size_t)L * L; c += blockDim.x) {
    const uint32_t i = (uint32_t)(c / L), j = (uint32_t)(c - (size_t)i * L);
    if (span && j < i) continue;  // the span layout holds the upper triangle only
    S[fold_sidx(span, L, W, i, j)] = j >= i + 3 ? w * (p[c] - th) - q[c] : 0.0f;  // a pair spans at least three (nussinov.cpp:236 is inside the span loop)
  }
}
__device__ void dd_fill_nw(uint32_t L1, uint32_t L2, uint32_t W, const float* __restrict__ p, const float* __restrict__ q, float* ps, float* qs) {
  for (size_t c = threadIdx.x; c < (size_t)L1 * L2; c += blockDim.x) {
    const uint32_t i = (uint32_t)(c / L2), k = (uint32_t)(c - (size_t)i * L2);
    const size_t o = nw_idx(L1, W, i + 1, k + 1);
    ps[o] = p[c];
    qs[o] = q[c];
  }
}


// ------------------------------------------------------------------------------------------
// standalone decoders (Fold::Decoder / Align::Decoder plugin calls)
// ------------------------------------------------------------------------------------------
// heads: candidates per column the workgroup form keeps in LDS (4 / 2 / 0), or DD_NONE when not even its rolling rows fit
// (beyond ~9 900 columns): the span-ordered form on global tables then
__global__ __launch_bounds__(DD_THREADS) void k_nussinov_single(uint32_t L, const float* p, const float* q, float w, float th,
                                                                nuss_ws ws, uint32_t* ss, float* score, uint32_t heads) {
  extern __shared__ unsigned char s_dd[];
  for (uint32_t i = threadIdx.x; i < L; i += blockDim.x) ss[i] = DD_NONE;
  if (heads != DD_NONE && L >= 3) {
    float* lds = (float*)(((uintptr_t)s_dd + 15) & ~(uintptr_t)15);
    const float sc = heads == 4 ? nuss_wg_span<4, true>(L, p, ws, lds, q, w, th) : heads == 2 ? nuss_wg_span<2, true>(L, p, ws, lds, q, w, th)
                                                                                              : nuss_wg_span<0, true>(L, p, ws, lds, q, w, th);
    if (threadIdx.x < 64) nuss_traceback_span(L, ws.tr, ss, (uint32_t*)lds, (int)threadIdx.x);
    if (threadIdx.x == 0) *score = sc;
    return;
  }
  nuss_ws none = {nullptr, nullptr, nullptr, nullptr, nullptr};
  nuss_pair_dp(L, p, q, w, ws, 0, nullptr, nullptr, 0.0f, none, th);
  if (threadIdx.x == 0) {
    nuss_traceback(L, ws, ss, ws.ck);  // the candidate-key array is free again: reuse it as the stack
    *score = ws.dp[L - 1];
  }
}

// Nussinov::decode, the dense class (reference src/nussinov.cpp:32-113 with q, :115-204 without): every pair scores
// sm = w(p-th)-q (or p-th), and the bifurcation runs over every split k in (i, j): dp[i][k] + dp[k+1][j].  Span-ordered,
// one barrier per span, a thread per cell; the traceback pushes (i,k) and (k+1,j) for code k-i+3.
__global__ __launch_bounds__(DD_THREADS) void k_nussinov_dense(uint32_t L, const float* p, const float* q, float w, float th,
                                                               float* dp, uint32_t* tr, uint32_t* stack, uint32_t* ss, float* score) {
  const uint32_t tid = threadIdx.x, nt = blockDim.x;
  for (size_t c = tid; c < (size_t)L * L; c += nt) { dp[c] = 0.0f; tr[c] = 0; }
  for (uint32_t i = tid; i < L; i += nt) ss[i] = DD_NONE;
  __syncthreads();
  for (uint32_t l = 1; l < L; ++l) {
    for (uint32_t i = tid; i + l < L; i += nt) {
      const uint32_t j = i + l;
      float v = 0.0f;
      uint32_t t = 0;
      if (i + 1 < j) { v = dp[(size_t)(i + 1) * L + j]; t = 1; }
      if (i < j - 1 && v < dp[(size_t)i * L + j - 1]) { v = dp[(size_t)i * L + j - 1]; t = 2; }
      const float pij = p[(size_t)i * L + j];
      const float sm = q ? w * (pij - th) - q[(size_t)i * L + j] : pij - th;
      if (i + 1 < j - 1) {
        const float c = dp[(size_t)(i + 1) * L + j - 1] + sm;
        if (v < c) { v = c; t = 3; }
      }
      for (uint32_t k = i + 1; k < j; ++k) {
        const float c = dp[(size_t)i * L + k] + dp[(size_t)(k + 1) * L + j];
        if (v < c) { v = c; t = k - i + 3; }
      }
      dp[(size_t)i * L + j] = v;
      tr[(size_t)i * L + j] = t;
    }
    __syncthreads();
  }
  if (tid == 0) {
    uint32_t sp = 0;
    stack[0] = 0; stack[1] = L - 1; sp = 1;
    uint32_t guard = 4 * L + 8;
    while (sp && guard--) {
      --sp;
      const int i = (int)stack[2 * sp], j = (int)stack[2 * sp + 1];
      const uint32_t t = tr[(size_t)i * L + j];
      if (t == 0) continue;
      if (t == 1) { stack[2 * sp] = i + 1; stack[2 * sp + 1] = j; ++sp; }
      else if (t == 2) { stack[2 * sp] = i; stack[2 * sp + 1] = j - 1; ++sp; }
      else if (t == 3) { ss[i] = j; stack[2 * sp] = i + 1; stack[2 * sp + 1] = j - 1; ++sp; }
      else {
        const int k = i + (int)t - 3;
        stack[2 * sp] = i; stack[2 * sp + 1] = k; ++sp;
        stack[2 * sp] = k + 1; stack[2 * sp + 1] = j; ++sp;
      }
    }
    *score = dp[L - 1];
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

// Profile averages (average_matching_probability dafs.cpp:513-559, average_basepairing_probability
// :561-607 without the alifold term).  Every cell of an averaged row receives at most one addend per
// source (a row of one sequence, or a pair of rows), and the reference adds them source by source, so:
// one wavefront per output row; the 64 lanes fetch the sparse rows of 64 sources at once (the pointer
// chases of 64 lookups overlap) and park the addends as (column, value) in LDS in source order; then
// every lane walks that list and applies the addends whose column it owns (column mod 64) to the row
// accumulator in LDS -- each cell sees its addends in the reference's order, no atomics.
#define AVG_STAGE 1024  // addends parked per round and wavefront
struct avg_src { const uint32_t* col; const float* val; uint32_t n; const uint32_t* map; };

template <class GetSrc>
__device__ void avg_row(uint32_t nsrc, float scale_div, GetSrc get, float* row, uint2* stage, int lane) {
  for (uint32_t base = 0; base < nsrc; base += 64) {
    const uint32_t me = base + lane;
    avg_src sr = {nullptr, nullptr, 0, nullptr};
    if (me < nsrc) sr = get(me);
    uint32_t first = 0;  // lanes < first have been applied
    while (first < 64) {
      // lanes first.. in order, as many as fit the staging area
      uint32_t cnt = (uint32_t)lane >= first ? sr.n : 0u;
      uint32_t incl = cnt;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const uint32_t up = __shfl_up(incl, o);
        if (lane >= o) incl += up;
      }
      const bool fits = incl <= AVG_STAGE;
      const unsigned long long fm = __ballot(fits);
      uint32_t last = first;
      {  // fits is monotone over lanes >= first: first lane that does not fit
        const unsigned long long nf = ~fm & ((~0ull) << first);
        last = nf ? (uint32_t)(__ffsll((long long)nf) - 1) : 64u;
      }
      if ((uint32_t)lane >= first && (uint32_t)lane < last) {
        const uint32_t pos = incl - cnt;
        for (uint32_t e = 0; e < sr.n; ++e) stage[pos + e] = make_uint2(sr.map[sr.col[e]], __float_as_uint(sr.val[e] / scale_div));
      }
      uint32_t total = __shfl(incl, (int)(last ? last - 1 : 0));
      if (last == first) {  // one source alone exceeds the stage (a row with more than AVG_STAGE entries): its lane applies it
        total = 0;
        if ((uint32_t)lane == first)
          for (uint32_t e = 0; e < sr.n; ++e) row[sr.map[sr.col[e]]] += sr.val[e] / scale_div;
        last = first + 1;
      }
      wave_lds_fence();
      for (uint32_t k = 0; k < total; ++k) {
        const uint2 a = stage[k];
        if ((a.x & 63u) == (uint32_t)lane) row[a.x] += __uint_as_float(a.y);
      }
      wave_lds_fence();
      first = last;
    }
  }
}

// coop != 0 (launches of few nodes whose p_z rows have hundreds of source rows each: the top of the guide tree): a
// workgroup per row of p_z.  A source is a chain of dependent loads (pair -> row pointers -> entries -> column map), and
// a wavefront on its own walks the n1*n2 sources of its row 64 at a time; here the four wavefronts fetch four batches
// at once, each into its own staging area, and the staged addends are then applied in batch order -- the order of the
// sources, r1 major -- by all threads (a thread owns the columns congruent to it).
__global__ __launch_bounds__(256) void k_node_avg(const dd_node* nodes, mp_store_dev mp, bp_store_dev bp, uint32_t row_cap, uint32_t coop) {
  extern __shared__ float s_rows[];  // (blockDim.x / 64) x row_cap: one accumulator row per wavefront
  __shared__ uint2 s_stage[4][AVG_STAGE];
  __shared__ uint32_t s_total[4];
  __shared__ uint32_t s_over;
  const dd_node nd = nodes[blockIdx.y];
  const uint32_t role = blockIdx.z;
  const int wave = (int)(threadIdx.x >> 6), lane = (int)(threadIdx.x & 63);
  if (coop && role == 2) {
    const uint32_t L1 = nd.L1, L2 = nd.L2, I = blockIdx.x, tid = threadIdx.x;
    if (I >= L1) return;
    const uint32_t nsrc = nd.n1 * nd.n2;
    const float nn = (float)nsrc;
    float* row = s_rows;
    for (uint32_t J = tid; J < L2; J += 256) row[J] = 0.0f;
    auto get = [&](uint32_t me) {
      avg_src sr = {nullptr, nullptr, 0, nullptr};
      const uint32_t r1 = me / nd.n2, r2 = me - r1 * nd.n2;
      const uint32_t ii = nd.rank1[(size_t)r1 * L1 + I];
      if (ii != DD_NONE) {
        const row_ref m = mp_row(mp, nd.seq1[r1], nd.seq2[r2], ii);
        sr.col = m.col; sr.val = m.val; sr.n = m.n; sr.map = nd.idx2 + nd.idxoff2[r2];
      }
      return sr;
    };
    __syncthreads();
    for (uint32_t base = 0; base < nsrc; base += 256) {
      if (tid == 0) s_over = 0;
      __syncthreads();
      {  // this wavefront's batch: sources base + 64 wave .. + 63, staged in source order
        const uint32_t me = base + 64u * (uint32_t)wave + (uint32_t)lane;
        avg_src sr = {nullptr, nullptr, 0, nullptr};
        if (me < nsrc) sr = get(me);
        uint32_t incl = sr.n;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
          const uint32_t up = __shfl_up(incl, o);
          if (lane >= o) incl += up;
        }
        const uint32_t total = __shfl(incl, 63);
        if (total <= AVG_STAGE) {
          const uint32_t pos = incl - sr.n;
          for (uint32_t e = 0; e < sr.n; ++e) s_stage[wave][pos + e] = make_uint2(sr.map[sr.col[e]], __float_as_uint(sr.val[e] / nn));
        } else if (lane == 0) s_over = 1;  // a batch that does not fit its staging area: the round goes the one-wavefront way
        if (lane == 0) s_total[wave] = total;
      }
      __syncthreads();
      if (s_over) {
        if (wave == 0) {
          const uint32_t hi = base + 256 < nsrc ? base + 256 : nsrc;
          avg_row(hi - base, nn, [&](uint32_t k) { return get(base + k); }, row, s_stage[0], lane);
        }
      } else {
        for (int w = 0; w < 4; ++w) {
          const uint32_t total = s_total[w];
          for (uint32_t k = 0; k < total; ++k) {
            const uint2 a = s_stage[w][k];
            if ((a.x & 255u) == tid) row[a.x] += __uint_as_float(a.y);
          }
        }
      }
      __syncthreads();
    }
    float* P = nd.p_z + (size_t)I * L2;
    for (uint32_t J = tid; J < L2; J += 256) {
      float v = row[J];
      if (v <= DD_CUTOFF) v = 0.0f;
      if (v > 1.0f) v = 1.0f;
      P[J] = v;
    }
    return;
  }
  const uint32_t I = blockIdx.x * (blockDim.x >> 6) + wave;
  float* row = s_rows + (size_t)wave * row_cap;
  uint2* stage = s_stage[wave];
  if (role < 2) {
    const uint32_t L = role ? nd.L2 : nd.L1, n = role ? nd.n2 : nd.n1;
    if (I >= L) return;
    const uint32_t* seq = role ? nd.seq2 : nd.seq1;
    const uint32_t* rank = role ? nd.rank2 : nd.rank1;
    const uint32_t* idx = role ? nd.idx2 : nd.idx1;
    const uint32_t* idxoff = role ? nd.idxoff2 : nd.idxoff1;
    float* P = (role ? nd.p_y : nd.p_x) + (size_t)I * L;
    for (uint32_t J = lane; J < L; J += 64) row[J] = 0.0f;
    wave_lds_fence();
    avg_row(n, (float)n, [&](uint32_t r) {
      avg_src sr = {nullptr, nullptr, 0, nullptr};
      const uint32_t ii = rank[(size_t)r * L + I];
      if (ii != DD_NONE) {
        const row_ref b = bp_row(bp, seq[r], ii);
        sr.col = b.col; sr.val = b.val; sr.n = b.n; sr.map = idx + idxoff[r];
      }
      return sr;
    }, row, stage, lane);
    for (uint32_t J = lane; J < L; J += 64) {
      float v = row[J];
      if (J > I && v <= DD_CUTOFF) v = 0.0f;
      P[J] = v;
    }
  } else {
    const uint32_t L1 = nd.L1, L2 = nd.L2;
    if (I >= L1) return;
    const uint32_t nn = nd.n1 * nd.n2;
    float* P = nd.p_z + (size_t)I * L2;
    for (uint32_t J = lane; J < L2; J += 64) row[J] = 0.0f;
    wave_lds_fence();
    for (uint32_t r1 = 0; r1 < nd.n1; ++r1) {
      const uint32_t ii = nd.rank1[(size_t)r1 * L1 + I];
      if (ii == DD_NONE) continue;
      const uint32_t s1 = nd.seq1[r1];
      avg_row(nd.n2, (float)nn, [&](uint32_t r2) {
        const row_ref m = mp_row(mp, s1, nd.seq2[r2], ii);
        avg_src sr = {m.col, m.val, m.n, nd.idx2 + nd.idxoff2[r2]};
        return sr;
      }, row, stage, lane);
    }
    for (uint32_t J = lane; J < L2; J += 64) {
      float v = row[J];
      if (v <= DD_CUTOFF) v = 0.0f;
      if (v > 1.0f) v = 1.0f;
      P[J] = v;
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

// In-place inclusive prefix sum of a[0..n) by the whole workgroup: every thread sums a contiguous
// chunk, the DD_THREADS chunk sums are scanned in LDS, every thread rewrites its chunk.  (A single
// thread walking a global array pays a memory round trip per element.)
__device__ void block_scan_inclusive(uint32_t* a, uint32_t n) {
  __shared__ uint32_t s_part[DD_THREADS];
  const uint32_t tid = threadIdx.x, nt = blockDim.x;
  const uint32_t chunk = (n + nt - 1) / nt;
  const uint32_t b = tid * chunk < n ? tid * chunk : n, e = b + chunk < n ? b + chunk : n;
  uint32_t sum = 0;
  for (uint32_t i = b; i < e; ++i) sum += a[i];
  s_part[tid] = sum;
  __syncthreads();
  if (tid < 64) {  // one wavefront scans the partials, eight per lane
    const uint32_t per = (nt + 63) / 64;
    uint32_t loc = 0;
    for (uint32_t k = 0; k < per; ++k) { const uint32_t id = tid * per + k; if (id < nt) loc += s_part[id]; }
    uint32_t incl = loc;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t up = __shfl_up(incl, o);
      if ((int)tid >= o) incl += up;
    }
    uint32_t run = incl - loc;
    for (uint32_t k = 0; k < per; ++k) {
      const uint32_t id = tid * per + k;
      if (id < nt) { const uint32_t v = s_part[id]; s_part[id] = run; run += v; }
    }
  }
  __syncthreads();
  uint32_t run = s_part[tid];
  for (uint32_t i = b; i < e; ++i) { run += a[i]; a[i] = run; }
  __syncthreads();
}

// sorted column lists of the entries > CUTOFF of every row (upper: only j > i).  A wavefront per row, the lanes
// across the columns (coalesced reads), positions by ballot
__device__ void row_lists(uint32_t R, uint32_t Cn, const float* P, bool upper, uint32_t* ptr, uint32_t* lst, int32_t* map) {
  const uint32_t tid = threadIdx.x, nt = blockDim.x;
  const uint32_t wave = tid >> 6, lane = tid & 63, nwaves = nt >> 6;
  // four 64-column chunks of a row are fetched before the first is counted: the walk is a chain of round trips to
  // L2 otherwise (a row of 150 columns: one trip instead of three)
  for (uint32_t i = wave; i < R; i += nwaves) {
    uint32_t c = 0;
    for (uint32_t j0 = upper ? ((i + 1) & ~63u) : 0; j0 < Cn; j0 += 256) {
      float v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const uint32_t j = j0 + 64 * u + lane;
        v[u] = (j < Cn && (!upper || j > i)) ? P[(size_t)i * Cn + j] : 0.0f;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) c += (uint32_t)__popcll(__ballot(v[u] > DD_CUTOFF));
    }
    if (lane == 0) ptr[i + 1] = c;
  }
  if (tid == 0) ptr[0] = 0;
  __syncthreads();
  block_scan_inclusive(ptr + 1, R);
  for (uint32_t i = wave; i < R; i += nwaves) {
    uint32_t pos = ptr[i];
    for (uint32_t j0 = upper ? ((i + 1) & ~63u) : 0; j0 < Cn; j0 += 256) {
      float v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const uint32_t j = j0 + 64 * u + lane;
        v[u] = (j < Cn && (!upper || j > i)) ? P[(size_t)i * Cn + j] : 0.0f;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const uint32_t j = j0 + 64 * u + lane;
        const bool keep = v[u] > DD_CUTOFF;
        const unsigned long long m = __ballot(keep);
        if (keep) {
          const uint32_t q = pos + (uint32_t)__popcll(m & ((1ull << lane) - 1));
          lst[q] = j;
          if (map) map[(size_t)i * Cn + j] = (int32_t)q;
        }
        pos += (uint32_t)__popcll(m);
      }
    }
  }
  __syncthreads();
}

// row of entry e: the largest i with ptr[i] <= e (ptr = exclusive row pointers, here an LDS copy)
__device__ __forceinline__ uint32_t row_of_entry(const uint32_t* ptr, uint32_t R, uint32_t e) {
  uint32_t lo = 0, hi = R;  // invariant: ptr[lo] <= e < ptr[hi]
  while (hi - lo > 1) {
    const uint32_t mid = (lo + hi) >> 1;
    if (ptr[mid] <= e) lo = mid; else hi = mid;
  }
  return lo;
}

// parts == 1: one workgroup does everything.  parts == 4 (launches of a few nodes, where the node's set-up stands between
// its children and its own first iteration): the lists of p_x, p_y, p_z and the alignment envelope + table initialisation
// are four workgroups (blockIdx.y); each raises the node's arrival counter (sync[5], zero since the node was carved) when
// its part is in memory, and the one that arrives last counts the consensus base pairs, which need all three lists.
__global__ __launch_bounds__(DD_THREADS) void k_node_lists(const dd_node* nodes, dd_params prm, uint32_t* ncbp_out, uint32_t pxptr_lds, uint32_t parts) {
  const dd_node nd = nodes[blockIdx.x];
  const uint32_t tid = threadIdx.x, nt = blockDim.x;
  const uint32_t L1 = nd.L1, L2 = nd.L2;
  const uint32_t part = blockIdx.y;
  if (parts == 1 || part == 0) row_lists(L1, L1, nd.p_x, true, nd.px_ptr, nd.px_j, nd.xmap);
  if (parts == 1 || part == 1) row_lists(L2, L2, nd.p_y, true, nd.py_ptr, nd.py_l, nd.ymap);
  if (parts == 1 || part == 2) row_lists(L1, L2, nd.p_z, false, nd.pz_ptr, nd.pz_k, nullptr);
  if (parts > 1) {
    if (part == 3) nw_envelope(L1, L2, nd.p_z, prm.th_a, nd.env, nd.x, nd.z);  // alignment envelope; x / z double as scratch here
    __shared__ uint32_t s_ticket;
    __threadfence();  // this part's lists, before the arrival
    __syncthreads();
    if (tid == 0) s_ticket = atomicAdd(&nd.sync[5], 1u);
    __syncthreads();
    if (s_ticket != parts - 1) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // the other parts' lists, after the arrival
    if (tid == 0) nd.sync[5] = 0;
  }
  // consensus base pairs per p_x entry (dafs.cpp:1022-1044)
  __shared__ uint32_t s_total;
  if (tid == 0) s_total = 0;
  __syncthreads();
  uint32_t mine = 0;
  // row pointers of p_x for the entry -> row searches: in LDS when the launch gave this node's L1 + 1 words room
  extern __shared__ uint32_t s_pxptr_dyn[];
  const uint32_t* s_pxptr = nd.px_ptr;
  if ((size_t)(L1 + 1) * 4 <= pxptr_lds) {
    for (uint32_t i = tid; i <= L1; i += nt) s_pxptr_dyn[i] = nd.px_ptr[i];
    s_pxptr = s_pxptr_dyn;
  }
  __syncthreads();
  const uint32_t npx = s_pxptr[L1];
  for (uint32_t e = tid; e < npx; e += nt) {  // a thread per entry of p_x
    {
      const uint32_t i = row_of_entry(s_pxptr, L1, e);
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
  }
  atomicAdd(&s_total, mine);
  if (parts == 1) nw_envelope(L1, L2, nd.p_z, prm.th_a, nd.env, nd.x, nd.z);  // alignment envelope; x / z double as scratch here
  __syncthreads();
  if (tid == 0 && ncbp_out) ncbp_out[blockIdx.x] = s_total;  // the host sizes the constraint block from this (one copy per call)
  if (tid == 0) { nd.info[0] = s_total; nd.info[1] = 0; nd.info[2] = 0; nd.info[3] = 0; nd.info[4] = 0; nd.info[5] = 0; nd.info[6] = 0; nd.info[7] = 0; }
}

// ---- set-up of WIDE nodes (round 3) ------------------------------------------------------------------------------------
// k_node_lists walks the dense matrices of a node with one workgroup (four in small launches): fine at a few hundred
// columns, 0.3 - 0.7 s per node where an alignment of 10 000 - 27 000 columns joins the tree (c5, random set: 4.5 s of the
// run in 164 launches).  Wide launches split the same work into kernels whose grids cover the ROWS of the three
// matrices: counts (+ the first / last column of the envelope from the same pass over p_z), a scan per matrix, the fill,
// the table initialisation -- and a last one-workgroup kernel for what is small (envelope smoothing, consensus-pair
// counts).  Same lists, same order, same envelope.
struct dd_mat { const float* P; uint32_t R, C; bool upper; uint32_t* ptr; uint32_t* lst; int32_t* map; };
__device__ __forceinline__ dd_mat dd_list_matrix(const dd_node& nd, uint32_t which) {
  if (which == 0) return dd_mat{nd.p_x, nd.L1, nd.L1, true, nd.px_ptr, nd.px_j, nd.xmap};
  if (which == 1) return dd_mat{nd.p_y, nd.L2, nd.L2, true, nd.py_ptr, nd.py_l, nd.ymap};
  return dd_mat{nd.p_z, nd.L1, nd.L2, false, nd.pz_ptr, nd.pz_k, nullptr};
}
// grid (row blocks of 8, node, matrix), 512 threads: a wavefront per row.  fill = 0: entries > CUTOFF per row -> ptr[i + 1];
// for p_z also the row's first / last column with p - th >= 0 (1-based, 0 = none) -> fa / la (= nd.x / nd.z, scratch
// here as in nw_envelope).  fill = 1: the columns (and the id map) at the scanned positions.
template <int WHICH>
__device__ __forceinline__ void lists_rows_body(const dd_node& nd, float th_a, int fill) {
  // (the matrix is a template argument: with the three cases merged behind one run-time selection the compiler left the
  // p_z case's pointers unset -- a fault at address 0 on the first launch; one instantiation per matrix is also leaner)
  const dd_mat m = dd_list_matrix(nd, (uint32_t)WHICH);
  const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const uint32_t i = blockIdx.x * 8 + wave;
  if (i >= m.R) return;
  const uint32_t jbeg = m.upper ? ((i + 1) & ~63u) : 0;
  const float* row = m.P + (size_t)i * m.C;
  if (!fill) {
    uint32_t c = 0, f = 0, l = 0;
    for (uint32_t j0 = jbeg; j0 < m.C; j0 += 256) {
      float v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const uint32_t j = j0 + 64 * u + lane;
        v[u] = (j < m.C && (!m.upper || j > i)) ? row[j] : (WHICH == 2 ? -1.0f : 0.0f);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        c += (uint32_t)__popcll(__ballot(v[u] > DD_CUTOFF));
        if (WHICH == 2) {  // needleman_wunsch.cpp:205-216: p - th >= 0 (padding lanes hold -1)
          const unsigned long long e = __ballot(v[u] - th_a >= 0.0f);
          if (e) {
            const uint32_t k0 = j0 + 64 * u;
            if (!f) f = k0 + (uint32_t)__ffsll((long long)e);
            l = k0 + 64 - (uint32_t)__clzll((long long)e);
          }
        }
      }
    }
    if (lane == 0) {
      m.ptr[i + 1] = c;
      if (WHICH == 2) { nd.x[i + 1] = f; nd.z[i + 1] = l; }
    }
    return;
  }
  uint32_t pos = m.ptr[i];
  for (uint32_t j0 = jbeg; j0 < m.C; j0 += 256) {
    float v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const uint32_t j = j0 + 64 * u + lane;
      v[u] = (j < m.C && (!m.upper || j > i)) ? row[j] : 0.0f;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const uint32_t j = j0 + 64 * u + lane;
      const bool keep = v[u] > DD_CUTOFF;
      const unsigned long long mk = __ballot(keep);
      if (keep) {
        const uint32_t q = pos + (uint32_t)__popcll(mk & ((1ull << lane) - 1));
        m.lst[q] = j;
        if (m.map) m.map[(size_t)i * m.C + j] = (int32_t)q;
      }
      pos += (uint32_t)__popcll(mk);
    }
  }
}
__global__ __launch_bounds__(512) void k_lists_rows(const dd_node* nodes, float th_a, int fill) {
  const dd_node nd = nodes[blockIdx.y];
  if (blockIdx.z == 0) lists_rows_body<0>(nd, th_a, fill);
  else if (blockIdx.z == 1) lists_rows_body<1>(nd, th_a, fill);
  else lists_rows_body<2>(nd, th_a, fill);
}
// grid (node, matrix): exclusive row pointers from the counts
__global__ __launch_bounds__(DD_THREADS) void k_lists_scan(const dd_node* nodes) {
  const dd_node nd = nodes[blockIdx.x];
  uint32_t* ptr = blockIdx.y == 0 ? nd.px_ptr : (blockIdx.y == 1 ? nd.py_ptr : nd.pz_ptr);
  const uint32_t R = blockIdx.y == 1 ? nd.L2 : nd.L1;
  if (threadIdx.x == 0) ptr[0] = 0;
  __syncthreads();
  block_scan_inclusive(ptr + 1, R);
}
// the envelope's sequential smoothing passes (needleman_wunsch.cpp:218-243) over first / last columns that are already
// there (fa = nd.x, la = nd.z): thread 0, on an LDS copy while it fits
__device__ void nw_envelope_smooth(uint32_t L1, uint32_t L2, const uint32_t* fa, const uint32_t* la, uint32_t* env) {
  __shared__ uint32_t s_env[2 * 4097];
  const uint32_t tid = threadIdx.x, nt = blockDim.x;
  const bool in_lds = L1 <= 4096;
  uint32_t* ev = in_lds ? s_env : env;
  if (tid == 0) {
    for (uint32_t i = 0; i <= L1; ++i) { ev[2 * i] = 0; ev[2 * i + 1] = 0; }
    for (uint32_t i = 1; i <= L1; ++i) {
      const uint32_t f = fa[i], l = la[i];
      if (f) {
        if (f - 1 < ev[2 * (i - 1)]) ev[2 * (i - 1)] = f - 1;
        ev[2 * i] = f;
      }
      if (ev[2 * i] == 0) {
        ev[2 * i] = ev[2 * (i - 1)];
        ev[2 * i + 1] = ev[2 * (i - 1) + 1];
        continue;
      }
      if (l - 1 > ev[2 * (i - 1) + 1]) ev[2 * (i - 1) + 1] = l - 1;
      ev[2 * i + 1] = l;
    }
    ev[2 * L1 + 1] = L2;
    for (uint32_t i = L1, v = L2; i != 0; --i) { v = v < ev[2 * i] ? v : ev[2 * i]; ev[2 * i] = v; }
    for (uint32_t i = 0, v = 0; i != L1 + 1; ++i) { v = v > ev[2 * i + 1] ? v : ev[2 * i + 1]; ev[2 * i + 1] = v; }
    for (uint32_t i = 1; i != L1 + 1; ++i)
      if (ev[2 * (i - 1) + 1] < ev[2 * i]) ev[2 * i] = ev[2 * (i - 1) + 1];
  }
  __syncthreads();
  if (in_lds) {
    for (uint32_t i = tid; i < 2 * (L1 + 1); i += nt) env[i] = s_env[i];
    __syncthreads();
  }
}
// grid (node): what is left of k_node_lists once the lists, the first / last columns and the table are there
__global__ __launch_bounds__(DD_THREADS) void k_node_lists_tail(const dd_node* nodes, dd_params prm, uint32_t* ncbp_out) {
  const dd_node nd = nodes[blockIdx.x];
  const uint32_t tid = threadIdx.x, nt = blockDim.x;
  const uint32_t L1 = nd.L1, L2 = nd.L2;
  __shared__ uint32_t s_total;
  if (tid == 0) s_total = 0;
  __syncthreads();
  uint32_t mine = 0;
  const uint32_t npx = nd.px_ptr[L1];
  for (uint32_t e = tid; e < npx; e += nt) {  // a thread per entry of p_x: consensus base pairs (dafs.cpp:1022-1044), as in k_node_lists
    const uint32_t i = row_of_entry(nd.px_ptr, L1, e);
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
  __syncthreads();
  nw_envelope_smooth(L1, L2, nd.x, nd.z, nd.env);
  if (tid == 0 && ncbp_out) ncbp_out[blockIdx.x] = s_total;
  if (tid == 0) { nd.info[0] = s_total; nd.info[1] = 0; nd.info[2] = 0; nd.info[3] = 0; nd.info[4] = 0; nd.info[5] = 0; nd.info[6] = 0; nd.info[7] = 0; }
}

__global__ __launch_bounds__(DD_THREADS) void k_node_cbp_fill(const dd_node* nodes, dd_params prm, uint32_t pxptr_lds) {
  const dd_node nd = nodes[blockIdx.x];
  const uint32_t tid = threadIdx.x, nt = blockDim.x;
  const uint32_t L1 = nd.L1, L2 = nd.L2;
  const uint32_t npx = nd.px_ptr[L1];
  if (nd.info[0] == 0) {
    // no consensus base pair (k_node_lists counted them): the flags keep their zero fill, the id map its -1, and the c_z
    // row lists are empty -- nothing to enumerate and no pass over the L1 x L2 cells (75 ms at the 27 000-column root of c5-random)
    for (uint32_t i = tid; i <= L1; i += nt) nd.cz_ptr[i] = 0;
    return;
  }
  // prefix of the per-entry counts (entries are already in (i,j) order); entry e starts at incl[e] - count[e],
  // i.e. at incl[e-1]
  block_scan_inclusive(nd.cbp_cnt, npx);
  extern __shared__ uint32_t s_pxptr_dyn[];
  const uint32_t* s_pxptr = nd.px_ptr;
  if ((size_t)(L1 + 1) * 4 <= pxptr_lds) {
    for (uint32_t i = tid; i <= L1; i += nt) s_pxptr_dyn[i] = nd.px_ptr[i];
    s_pxptr = s_pxptr_dyn;
  }
  __syncthreads();
  for (uint32_t e = tid; e < npx; e += nt) {  // a thread per entry of p_x
    {
      const uint32_t i = row_of_entry(s_pxptr, L1, e);
      const uint32_t j = nd.px_j[e];
      const float px = nd.p_x[(size_t)i * L1 + j];
      uint32_t u = e ? nd.cbp_cnt[e - 1] : 0u;
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
  }
  __syncthreads();
  // c_z as sorted row lists (:1056-1060) + dense id map
  const uint32_t wave = tid >> 6, lane = tid & 63, nwaves = nt >> 6;
  for (uint32_t i = wave; i < L1; i += nwaves) {
    uint32_t c = 0;
    for (uint32_t k0 = 0; k0 < L2; k0 += 64) {
      const uint32_t k = k0 + lane;
      c += (uint32_t)__popcll(__ballot(k < L2 && nd.cz_flag[(size_t)i * L2 + k] != 0));
    }
    if (lane == 0) nd.cz_ptr[i + 1] = c;
  }
  if (tid == 0) nd.cz_ptr[0] = 0;
  __syncthreads();
  block_scan_inclusive(nd.cz_ptr + 1, L1);
  for (uint32_t i = wave; i < L1; i += nwaves) {
    uint32_t pos = nd.cz_ptr[i];
    for (uint32_t k0 = 0; k0 < L2; k0 += 64) {
      const uint32_t k = k0 + lane;
      const bool keep = k < L2 && nd.cz_flag[(size_t)i * L2 + k] != 0;
      const unsigned long long m = __ballot(keep);
      if (keep) {
        const uint32_t q = pos + (uint32_t)__popcll(m & ((1ull << lane) - 1));
        nd.cz_k[q] = k;
        nd.zmap[(size_t)i * L2 + k] = (int32_t)q;
      }
      pos += (uint32_t)__popcll(m);
    }
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
// split mode: a folding DP on a workgroup of its own.  The leader (k_dd_solve, blockIdx.y == 0) publishes
// the iteration to run in sync[0] after the multiplier updates of the previous one are visible; the folder
// runs DP + traceback, publishes the score and raises its counter.  All waits are bounded.
// ------------------------------------------------------------------------------------------
#define DD_SYNC_EXIT 0xFFFFFFFFu
// How long one side of a split node waits for the other before it takes it for lost, in wall_clock64 ticks (100 MHz):
// two seconds plus 40 ns per cell of the wider folding -- ten times what a pass of the span-ordered form on HBM tables
// costs (6.5 ms at 1100 columns, 26 ms at 2650: ~3.7 ns per cell), so that the folders of the widest nodes (11 500 columns:
// ~0.5 s a pass) are not declared lost while they work.
__device__ __forceinline__ unsigned long long dd_lost_ticks(uint32_t L1, uint32_t L2) {
  const unsigned long long L = L1 > L2 ? L1 : L2;
  return 200000000ull + 4ull * L * L;
}
__device__ __forceinline__ uint32_t sync_load(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void sync_store(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ void dd_folder(const dd_node& nd, const dd_params& prm, uint32_t role, uint32_t t_first) {
  extern __shared__ unsigned char s_dd[];
  __shared__ float s_fscore;
  __shared__ uint32_t s_go, s_slow;
  bool gave_up = false;  // the register form overflowed earlier in this launch: straight to the span-ordered form
  const uint32_t tid = threadIdx.x, nt = blockDim.x;
  const int wave = (int)(tid >> 6), lane = (int)(tid & 63);
  const bool isx = role == 1;
  const uint32_t L = isx ? nd.L1 : nd.L2;
  const uint32_t W = dd_fold_cols(L);
  const float* S = isx ? nd.s_x : nd.s_y;
  const nuss_ws& ws = isx ? nd.wx : nd.wy;
  uint8_t* trb_g = isx ? nd.trb_x : nd.trb_y;
  uint32_t* ss = isx ? nd.x : nd.y;
  const uint32_t nw = (uint32_t)(((size_t)L * (L + 1) / 2 + 7) / 8);
  uint32_t *trbp = nullptr, *lck = nullptr;
  float* ring = nullptr;
  float* P = (float*)s_dd;  // traceback stack of the register form: inside its ring, which is idle by then
  // span form (fold_fast bit 4 / 5): codes, dp triangle, candidate values and row offsets, split rows -- as in k_dd_solve
  const bool span = (nd.fold_fast & (isx ? 16u : 32u)) != 0;
  const bool span_mw = L > 64 && !prm.span_one_wave;  // a wavefront per row slot (one slot: nothing to share out)
  float *tri = nullptr, *cvl = nullptr;
  uint32_t* ckl = nullptr;
  if (span) {
    uint32_t* w = (uint32_t*)(((uintptr_t)s_dd + 15) & ~(uintptr_t)15);
    trbp = w; w += dd_span_nib_words(L); tri = (float*)w; w += dd_span_tri_words(L);
    cvl = (float*)w; w += DD_CAP * (L + 1); ckl = w; w += DD_CAP * (L + 1); lck = w;
    P = cvl;  // traceback stack: the candidate values are dead once the DP is through
    for (uint32_t e = tid; e < dd_span_tri_words(L); e += nt) tri[e] = 0.0f;  // spans 0..2 hold 0 and are never written
  } else
  if (nd.fold_fast & (isx ? 1u : 2u)) { trbp = (uint32_t*)s_dd; ring = (float*)(trbp + nw); lck = (uint32_t*)(ring + dd_ring_words(L)); P = ring; }
  else if (nd.fold_fast & (isx ? 4u : 8u)) { ring = (float*)s_dd; lck = (uint32_t*)(ring + dd_ring_words(L)); P = ring; }  // codes in HBM
  for (uint32_t it = t_first;; ++it) {
    if (tid == 0) {
      uint32_t g = 0;
      const unsigned long long t_wait = wall_clock64(), t_max_wait = dd_lost_ticks(nd.L1, nd.L2);
      while ((g = sync_load(&nd.sync[0])) != it + 1 && g != DD_SYNC_EXIT && wall_clock64() - t_wait <= t_max_wait) __builtin_amdgcn_s_sleep(16);
      s_go = (g == it + 1) ? 1u : 0u;  // a leader that does not show up in time: leave (it will find its folders lost and be relaunched)
    }
    __syncthreads();
    if (!s_go) break;
    for (uint32_t i = tid; i < L; i += nt) ss[i] = DD_NONE;
    if (trbp) for (uint32_t e = tid; e < nw; e += nt) trbp[e] = 0;
    if (span) for (uint32_t e = tid; e < DD_CAP * (L + 1); e += nt) { cvl[e] = -INFINITY; ckl[e] = 0; }  // empty candidate lists
    __syncthreads();
    // The register form first, unless there is none for this width (beyond 768 columns, or no room) or it has
    // already overflowed its DD_CAP candidates per column in this launch (dense inputs do so every time).
    if (span && !gave_up && span_mw) {
      // a wavefront per row slot, a barrier per span (nuss_span_mw); the traceback stays with wave 0
      if (tid == 0) s_slow = 0;
      __syncthreads();
      bool slow = true;
      const unsigned long long tf0 = prm.stamps ? wall_clock64() : 0ull;
      const float sc = nuss_span_mw(L, isx ? nd.s_xs : nd.s_ys, trbp, tri, cvl, ckl, lck, wave, lane, &slow);
      if (slow && lane == 0) atomicOr(&s_slow, 1u);
      __syncthreads();
      if (wave == 0) {
        const unsigned long long tf1 = prm.stamps ? wall_clock64() : 0ull;
        if (!s_slow) nuss_traceback_fast(L, trbp, trb_g, lck, ss, (uint32_t*)P, lane);
        if (lane == 0) {
          s_fscore = sc;
          if (prm.stamps) { nd.sync[isx ? 5 : 6] += (uint32_t)(tf1 - tf0); nd.sync[7] += (uint32_t)(wall_clock64() - tf1); }  // DP of x / y, tracebacks of both
        }
      }
      __syncthreads();
      gave_up = s_slow != 0;
    } else
    if ((span || (ring && W <= DD_WFOLD)) && !gave_up) {
      if (wave == 0) {
        bool slow = true;
        const float sc = span ? nuss_wave_span_t(L, isx ? nd.s_xs : nd.s_ys, trbp, tri, cvl, ckl, lck, lane, &slow)
                              : nuss_wave_fast(W, L, S, trbp, trb_g, ring, lck, lane, &slow);
        if (!slow) nuss_traceback_fast(L, trbp, trb_g, lck, ss, (uint32_t*)P, lane);
        if (lane == 0) { s_fscore = sc; s_slow = slow ? 1u : 0u; }
      }
      __syncthreads();
      gave_up = s_slow != 0;
    } else gave_up = true;
    if (gave_up) {
      // The span-ordered form on all threads of the workgroup, one barrier per span -- the standalone decoder's DP,
      // which outruns the HBM-table wave form from a few hundred columns on (2143 columns: 31 -> 11 ms a pass).
      // It takes p and q as they are (no sweep-order copy).
      const nuss_ws none = {nullptr, nullptr, nullptr, nullptr, nullptr};
      const float wf = prm.w * 2 * (isx ? nd.n1 : nd.n2) / (nd.n1 + nd.n2);  // dafs.cpp:1091-1092, as in k_dd_solve
      for (uint32_t i = tid; i < L; i += nt) ss[i] = DD_NONE;  // a register-form traceback cut short may have left marks
      if (nd.fold_fast & (isx ? 64u : 128u)) {
        // beyond the register forms: the workgroup form (rolling rows and candidate heads in LDS, tables by span)
        float* lds = (float*)(((uintptr_t)s_dd + 15) & ~(uintptr_t)15);
        const uint32_t K = (nd.fold_fast >> (isx ? 8 : 12)) & 15u;
        const float* Ss = isx ? nd.s_xs : nd.s_ys;
        const unsigned long long tf0 = prm.stamps ? wall_clock64() : 0ull;
        const float sc = K == 4 ? nuss_wg_span<4>(L, Ss, ws, lds) : K == 2 ? nuss_wg_span<2>(L, Ss, ws, lds) : nuss_wg_span<0>(L, Ss, ws, lds);
        if (wave == 0) {
          const unsigned long long tf1 = prm.stamps ? wall_clock64() : 0ull;
          nuss_traceback_span(L, ws.tr, ss, (uint32_t*)lds, lane);  // the rolling rows are free again: the stack
          if (lane == 0) {
            s_fscore = sc;
            if (prm.stamps) { nd.sync[isx ? 5 : 6] += (uint32_t)(tf1 - tf0); nd.sync[7] += (uint32_t)(wall_clock64() - tf1); }  // DP of x / y, tracebacks of both
          }
        }
      } else {
      nuss_pair_dp(L, isx ? nd.p_x : nd.p_y, isx ? nd.q_x : nd.q_y, wf, ws, 0, nullptr, nullptr, 0.0f, none, prm.th_s);
      if (tid == 0) {
        nuss_traceback(L, ws, ss, ws.ck);  // the candidate-key array is free again: reuse it as the stack
        s_fscore = ws.dp[L - 1];
      }
      }
    }
    __syncthreads();
    if (tid == 0) {
      nd.sync[2 + role] = __float_as_uint(s_fscore);
      sync_store(&nd.sync[role], it + 1);
    }
  }
}

// ------------------------------------------------------------------------------------------
// the subgradient loop, dafs.cpp:1066-1294
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(DD_SOLVE_THREADS) void k_dd_solve(const dd_node* nodes, dd_params prm, uint32_t* paused_out) {
  // The node descriptor (sixty-odd pointers) lives in LDS: as a private copy it went to scratch memory (568 bytes per
  // lane, re-read from there a hundred times in the loop below), because the out-of-line helpers take it by reference.
  __shared__ dd_node s_nd;
  {
    const uint32_t* src = (const uint32_t*)(nodes + blockIdx.x);
    uint32_t* dst = (uint32_t*)&s_nd;
    for (uint32_t k = threadIdx.x; k < sizeof(dd_node) / 4; k += blockDim.x) dst[k] = src[k];
  }
  __syncthreads();
  const dd_node& nd = s_nd;
  if (blockIdx.y != 0) {  // folding workgroups of a split node
    if (nd.split) dd_folder(nd, prm, blockIdx.y, nd.info[6] != 0 ? nd.info[1] : 0u);
    return;
  }
  const bool split = nd.split != 0;
  const uint32_t tid = threadIdx.x, nt = blockDim.x;
  const uint32_t L1 = nd.L1, L2 = nd.L2;
  const uint32_t ncbp = nd.info[0];
  const uint32_t npx = nd.px_ptr[L1], npy = nd.py_ptr[L2], ncz = nd.cz_ptr[L1];
  const float w_x = prm.w * 2 * nd.n1 / (nd.n1 + nd.n2);  // dafs.cpp:1091
  const float w_y = prm.w * 2 * nd.n2 / (nd.n1 + nd.n2);  // :1092
  __shared__ uint32_t s_violated, s_npos;
  __shared__ int s_stop, s_bad, s_lost;  // s_lost: a folder of this split node did not answer in time
  __shared__ float s_eta;
  __shared__ float s_score[3];
  __shared__ uint32_t s_slowxy[2];       // this iteration's x / y folding is still to be done by the span-ordered form
  bool gave_up_x = false, gave_up_y = false;
  float c = 0.0f, eta = prm.eta0, s_prev = 0.0f;  // meaningful in thread 0
  uint32_t t = 0, violated = 0;
  const bool resume = nd.info[6] != 0;  // a node paused by an earlier launch (prm.slice): pick up its loop state
  const uint32_t t_first = resume ? nd.info[1] : 0;
  if (tid == 0) {
    if (resume) { c = nd.fstate[0]; eta = nd.fstate[1]; s_prev = nd.fstate[2]; }
    s_eta = eta; s_bad = 0; s_lost = 0;
  }
  // dynamic LDS: whichever traceback tables and in-flight rows fit (nd.lds_flags, decided by the host): bit 0 alignment,
  // bit 1 x, bit 2 y
  extern __shared__ unsigned char s_dd[];
  const uint32_t Wx = dd_fold_cols(L1), Wy = dd_fold_cols(L2), Wz = nd.nw_w;
  // previous-row buffers and candidate counters of the HBM-table folding forms: room of their own only when the
  // fold has no on-chip region (otherwise they borrow its ring, see dd_ring_words) and this workgroup folds at all
  const bool fastx = (nd.lds_flags & (2u | 8u)) != 0, fasty = (nd.lds_flags & (4u | 8u)) != 0;
  float *Px = nullptr, *Py = nullptr;
  unsigned char* lds_tail;
  lds_tail = s_dd;
  // bit 0: packed alignment traceback; bit 1 / bit 2: the fast form of the x / y folding DP
  // (in-flight rows, candidate lists and packed traceback codes)
  const uint32_t nzw = dd_nwtab_words(L1, L2), nxw = (uint32_t)(((size_t)L1 * (L1 + 1) / 2 + 7) / 8),
                 nyw = (uint32_t)(((size_t)L2 * (L2 + 1) / 2 + 7) / 8);
  uint32_t *trzp = nullptr, *trxp = nullptr, *tryp = nullptr;
  float *ringx = nullptr, *ringy = nullptr;
  uint32_t *lckx = nullptr, *lcky = nullptr;
  // span form (bit 6): dp triangle, candidate values and row offsets of each folding
  const bool spanxy = (nd.lds_flags & 64u) != 0;
  float *trix = nullptr, *triy = nullptr, *cvx = nullptr, *cvy = nullptr;
  uint32_t *ckx = nullptr, *cky = nullptr;
  {
    uint32_t* w = (uint32_t*)lds_tail;
    if (spanxy) {
      w = (uint32_t*)(((uintptr_t)w + 15) & ~(uintptr_t)15);  // 16-byte candidate slots
      trxp = w; w += dd_span_nib_words(L1); trix = (float*)w; w += dd_span_tri_words(L1);
      cvx = (float*)w; w += DD_CAP * (L1 + 1); ckx = w; w += DD_CAP * (L1 + 1); lckx = w; w += DD_CAP * L1;
      tryp = w; w += dd_span_nib_words(L2); triy = (float*)w; w += dd_span_tri_words(L2);
      cvy = (float*)w; w += DD_CAP * (L2 + 1); cky = w; w += DD_CAP * (L2 + 1); lcky = w; w += DD_CAP * L2;
    }
    if (nd.lds_flags & 1) { trzp = w; w += nzw; }
    if (nd.lds_flags & 2) { trxp = w; w += nxw; ringx = (float*)w; w += dd_ring_words(L1); lckx = w; w += DD_CAP * L1; }
    if (nd.lds_flags & 4) { tryp = w; w += nyw; ringy = (float*)w; w += dd_ring_words(L2); lcky = w; w += DD_CAP * L2; }
    if (nd.lds_flags & 8) {  // one region for both folding DPs, used by x and then by y
      const uint32_t Lm = L1 > L2 ? L1 : L2, rw1 = dd_ring_words(L1), rw2 = dd_ring_words(L2);
      if (!(nd.lds_flags & 16)) { trxp = tryp = w; w += nxw > nyw ? nxw : nyw; }  // bit 4: the codes go to HBM instead
      ringx = ringy = (float*)w; w += rw1 > rw2 ? rw1 : rw2;
      lckx = lcky = w; w += DD_CAP * Lm;
    }
    if (fastx) Px = ringx;
    if (fasty) Py = ringy;
    if (spanxy) { Px = cvx; Py = cvy; }  // traceback stacks: the candidate values are dead once the DP is through
  }

  const bool shared_xy = (nd.lds_flags & 8) != 0;
  __shared__ uint32_t s_x_done;  // iteration whose x folding (DP + traceback) has released the shared region
  if (tid == 0) s_x_done = 0xFFFFFFFFu;
  uint8_t* trz = nd.tr_z;
  if (spanxy) {  // spans 0..2 of the dp triangles hold 0 and are never written
    for (uint32_t e = tid; e < dd_span_tri_words(L1); e += nt) trix[e] = 0.0f;
    for (uint32_t e = tid; e < dd_span_tri_words(L2); e += nt) triy[e] = 0.0f;
  }
  if (!resume) {
    // sweep-order inputs of the three DPs, built once; the multiplier updates below keep them current
    if (!(prm.skip_xy && ncbp == 0)) {  // a node that leaves its foldings out (see fold_on below) needs no scores
      if (nd.s_x) dd_fill_scores(false, L1, nd.p_x, nd.q_x, w_x, prm.th_s, nd.s_x);  // only foldings with a register form keep one
      if (nd.s_y) dd_fill_scores(false, L2, nd.p_y, nd.q_y, w_y, prm.th_s, nd.s_y);
      if (nd.s_xs) dd_fill_scores(true, L1, nd.p_x, nd.q_x, w_x, prm.th_s, nd.s_xs);  // the copy the span form reads
      if (nd.s_ys) dd_fill_scores(true, L2, nd.p_y, nd.q_y, w_y, prm.th_s, nd.s_ys);
    }
    dd_fill_nw(L1, L2, Wz, nd.p_z, nd.q_z, nd.pz_s, nd.qz_s);
    for (uint32_t r = tid; r < L1 + 130; r += nt) {  // the padded envelope (dd_node::env4)
      const bool in = r >= 65 && r <= L1 + 64;        // rows 1 .. L1
      const uint32_t ef = in ? nd.env[2 * (r - 64)] : 1u, es = in ? nd.env[2 * (r - 64) + 1] : 0u;
      nd.env4[2 * r] = in ? (ef > 1u ? ef : 1u) : 1u;
      nd.env4[2 * r + 1] = es;
    }
  }
  __syncthreads();
  const int wave = (int)(tid >> 6), lane = (int)(tid & 63);

  // optional phase timing (100 MHz ticks accumulated over the iterations into info[8..13]; tuning aid)
  unsigned long long tk[7] = {0, 0, 0, 0, 0, 0, 0}, t_prev = 0;
  __shared__ unsigned long long s_tky, s_tkz;  // time of the y folding and of the alignment DP, by their own wavefronts
  if (tid == 0) { s_tky = 0; s_tkz = 0; }
#define DD_TICK(k) if (prm.stamps && tid == 0) { const unsigned long long now = wall_clock64(); tk[k] += now - t_prev; t_prev = now; }
  if (prm.stamps && tid == 0) t_prev = wall_clock64();
  // time bound of the launch (dd_params::budget); meaningful in thread 0
  unsigned long long deadline = 0;
  if (prm.budget && tid == 0) {
    unsigned long long t0 = wall_clock64();
    if (prm.t_ref) {
      if (prm.t_ref_write) {
        if (blockIdx.x == 0) __hip_atomic_store(prm.t_ref, t0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        const unsigned long long ref = __hip_atomic_load(prm.t_ref, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (ref) t0 = ref;
      }
    }
    deadline = t0 + prm.budget;
  }
  uint32_t ran = 0;
  bool paused = false;
  // prm.skip_xy: a node without consensus base pairs has nothing that couples its three subproblems, and the
  // caller has said it consumes the alignment alone (DAFS::align_alignments, dafs.cpp:896-912): no folding DPs
  const bool fold_on = !(prm.skip_xy && ncbp == 0);
  if (split && !fold_on && tid == 0) sync_store(&nd.sync[0], DD_SYNC_EXIT);  // nothing to fold: the two folding workgroups leave at once
  for (t = t_first; t != prm.t_max; ++t) {
    if (split && fold_on) {
      // every thread's multiplier updates are out (the barrier that ended the previous iteration, or the one
      // after the initial fill); one release publishes them together with the go signal
      if (tid == 0) sync_store(&nd.sync[0], t + 1);
    } else {
      for (uint32_t i = tid; i < L1; i += nt) nd.x[i] = DD_NONE;
      for (uint32_t k = tid; k < L2; k += nt) nd.y[k] = DD_NONE;
    }
    if (tid == 0) { s_slowxy[0] = 0; s_slowxy[1] = 0; }
    // packed traceback tables are filled by OR
    if (trzp) for (uint32_t e = tid; e < nzw; e += nt) trzp[e] = 0;
    if (trxp) for (uint32_t e = tid; e < nxw; e += nt) trxp[e] = 0;
    if (tryp && !shared_xy) for (uint32_t e = tid; e < nyw; e += nt) tryp[e] = 0;
    if (spanxy) {  // empty candidate lists (entry L / L2 is the list that stays empty)
      for (uint32_t e = tid; e < DD_CAP * (L1 + 1); e += nt) { cvx[e] = -INFINITY; ckx[e] = 0; }
      for (uint32_t e = tid; e < DD_CAP * (L2 + 1); e += nt) { cvy[e] = -INFINITY; cky[e] = 0; }
    }
    // the three subproblems (dafs.cpp:1091-1093) side by side, one wavefront each, DP then traceback
    __syncthreads();
    if (!fold_on) {
      if (tid == 0) { s_score[0] = 0.0f; s_score[1] = 0.0f; }
    } else if (split) {
      // the folders are at work on their own CUs
    } else if (wave == 0) {
      bool slow = true;
      float sc = 0.0f;
      if (spanxy && !gave_up_x) sc = nuss_wave_span_t(L1, nd.s_xs, trxp, trix, cvx, ckx, lckx, lane, &slow);
      else if (ringx && !gave_up_x && Wx <= DD_WFOLD) sc = nuss_wave_fast(Wx, L1, nd.s_x, trxp, nd.trb_x, ringx, lckx, lane, &slow);
      if (slow && lane == 0 && prm.stamps) nd.info[4] += 1;  // iterations that took the slower form
      if (lane == 0) { s_slowxy[0] = slow ? 1u : 0u; s_score[0] = sc; }  // slow: the span-ordered form below, by everybody
      DD_TICK(0);
      if (!slow) nuss_traceback_fast(L1, trxp, nd.trb_x, lckx, nd.x, (uint32_t*)Px, lane);
      DD_TICK(1);
      if (shared_xy) {  // hand the region to the y folding
        wave_lds_fence();
        if (lane == 0) __hip_atomic_store(&s_x_done, t, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    } else if (wave == 1) {
      bool slow = true;
      float sc = 0.0f;
      const unsigned long long ty0 = prm.stamps ? wall_clock64() : 0ull;
      if (shared_xy) {
        while (__hip_atomic_load(&s_x_done, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != t) __builtin_amdgcn_s_sleep(8);
        if (tryp) for (uint32_t e = (uint32_t)lane; e < nyw; e += 64) tryp[e] = 0;
        wave_lds_fence();
      }
      if (spanxy && !gave_up_y) sc = nuss_wave_span_t(L2, nd.s_ys, tryp, triy, cvy, cky, lcky, lane, &slow);
      else if (ringy && !gave_up_y && Wy <= DD_WFOLD) sc = nuss_wave_fast(Wy, L2, nd.s_y, tryp, nd.trb_y, ringy, lcky, lane, &slow);
      if (slow && lane == 0 && prm.stamps) nd.info[5] += 1;
      if (lane == 0) { s_slowxy[1] = slow ? 1u : 0u; s_score[1] = sc; }
      if (!slow) nuss_traceback_fast(L2, tryp, nd.trb_y, lcky, nd.y, (uint32_t*)Py, lane);
      if (prm.stamps && lane == 0) s_tky += wall_clock64() - ty0;
    }
    if (wave == 2) {
      float sc;
      const unsigned long long tz0 = prm.stamps ? wall_clock64() : 0ull;
      // the register form: codes packed in LDS when the plan has room for them (bit 0; one panel, up to DD_WNW columns per
      // lane), else in HBM slots (any width: panels of 64 Wz columns)
      const bool reg_l = trzp && Wz <= DD_WNW && L2 < 64u * Wz;
      if (reg_l) sc = nw_wave_fast<1>(Wz, L1, L2, nd.pz_s, nd.qz_s, prm.th_a, nd.env4, (uint8_t*)trzp, nullptr, lane);
      else sc = nw_wave_fast<2>(Wz, L1, L2, nd.pz_s, nd.qz_s, prm.th_a, nd.env4, trz, nd.nw_edge, lane);
      bool ok = true;
      if (reg_l) { wave_lds_fence(); ok = nw_traceback_wave<false>(L1, L2, trzp, nd.z, lane); }  // by the whole wavefront
      else { __threadfence_block(); ok = nw_traceback_wave<true>(L1, L2, (const uint32_t*)trz, nd.z, lane, Wz); }
      if (lane == 0) {
        s_score[2] = sc;
        if (!ok) s_bad = 1;
        if (prm.stamps) s_tkz += wall_clock64() - tz0;
      }
    }
    DD_TICK(0);
    if (tid >= 192) {
      // The fourth wavefront, beside the three DPs: the consensus constraints (:1103-1117).  s_w of a consensus pair is a
      // sum of multipliers as the previous iteration left them -- nothing of this iteration's subproblems enters -- so
      // the counts (atomics) and the list of positive s_w (compacted in consensus-pair order: the dual value adds them in
      // that order) are ready when the DPs are.
      const uint32_t l3 = tid - 192;
      for (uint32_t e = l3; e < npx; e += 64) nd.tx[e] = 0;
      for (uint32_t e = l3; e < npy; e += 64) nd.ty[e] = 0;
      for (uint32_t e = l3; e < ncz; e += 64) nd.tz[e] = 0;
      __threadfence();  // the zeros are in place before the first count
      const uint32_t chunk = (ncbp + 63) / 64;
      const uint32_t u0 = l3 * chunk < ncbp ? l3 * chunk : ncbp;
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
      uint32_t incl = npos;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const uint32_t up = __shfl_up(incl, o);
        if ((int)l3 >= o) incl += up;
      }
      if (l3 == 63) s_npos = incl;
      uint32_t pos = incl - npos;
      for (uint32_t u = u0; u < u1; ++u) {
        const uint32_t* cb = nd.cbp + (size_t)8 * u;
        const float s_w = nd.q_x[(size_t)cb[0] * L1 + cb[1]] + nd.q_y[(size_t)cb[2] * L2 + cb[3]] -
                          nd.q_z[(size_t)cb[0] * L2 + cb[2]] - nd.q_z[(size_t)cb[1] * L2 + cb[3]];
        if (s_w > 0.0f) nd.sw[pos++] = s_w;
      }
    }
    if (tid == 0) {
      s_violated = 0;
      if (split && fold_on) {
        // Collect the two foldings.  The wait is bounded in time (wall_clock64, 100 MHz), not in spins: split mode only
        // works while the node's three workgroups are on the machine together, and a kernel on another stream, a second
        // context or a masked device can keep the folders off it.  A folder that has not answered after dd_lost_ticks (two
        // seconds + ten times the cost of its pass) is taken for lost: nothing of iteration t has been applied yet (the multiplier updates follow), so the node is
        // parked at t like a node whose slice ran out, marked for the one-workgroup form, and the host relaunches it.
        const unsigned long long t_wait = wall_clock64(), t_max_wait = dd_lost_ticks(L1, L2);
        bool lost = prm.debug_lose_folders != 0;  // tests: take the recovery path without waiting
        while (!lost && (sync_load(&nd.sync[1]) != t + 1 || sync_load(&nd.sync[2]) != t + 1)) {
          __builtin_amdgcn_s_sleep(16);
          if (wall_clock64() - t_wait > t_max_wait) lost = true;
        }
        if (lost) s_lost = 1;
        s_score[0] = __uint_as_float(nd.sync[3]);
        s_score[1] = __uint_as_float(nd.sync[4]);
      }
    }
    __syncthreads();
    if (s_lost) { paused = true; break; }  // every thread sees it after the barrier; iteration t is redone by the relaunch
    {
      // Foldings whose register form gave up (more than DD_CAP candidates in a column) or does not exist for this
      // width: the span-ordered form of the standalone decoder, both at once, on all threads; and no further
      // register attempts in this launch.
      const bool slx = s_slowxy[0] != 0, sly = s_slowxy[1] != 0;
      gave_up_x = gave_up_x || slx;
      gave_up_y = gave_up_y || sly;
      if (slx || sly) {
        nuss_pair_dp(slx ? L1 : 0u, nd.p_x, nd.q_x, w_x, nd.wx, sly ? L2 : 0u, nd.p_y, nd.q_y, w_y, nd.wy, prm.th_s);
        if (tid == 0 && slx) { nuss_traceback(L1, nd.wx, nd.x, nd.wx.ck); s_score[0] = nd.wx.dp[L1 - 1]; }
        if (tid == 64 && sly) { nuss_traceback(L2, nd.wy, nd.y, nd.wy.ck); s_score[1] = nd.wy.dp[L2 - 1]; }
        __syncthreads();
      }
    }
    DD_TICK(2);

    DD_TICK(3);
    eta = s_eta;

    // multiplier updates (:1121-1254), every cell touched by exactly one lane
    uint32_t viol = 0;
    for (uint32_t i = tid; i < L1; i += nt) {
      // The loads of a row's x and z cell are issued level by level, whether the row has such a cell or not (the indices of
      // the missing ones point at cell 0 of the row): the chain x -> map -> count -> multiplier, followed by the same chain
      // for z, was a dozen dependent trips to L2 per iteration.
      const uint32_t j = nd.x[i], kz = nd.z[i];
      const uint32_t pe0 = nd.px_ptr[i], pe1 = nd.px_ptr[i + 1], ce0 = nd.cz_ptr[i], ce1 = nd.cz_ptr[i + 1];
      const size_t ox = (size_t)i * L1 + (j != DD_NONE ? j : 0u), oz = (size_t)i * L2 + (kz != DD_NONE ? kz : 0u);
      const int32_t idx = nd.xmap[ox], idz = nd.zmap[oz];
      const float qx0 = nd.q_x[ox], px0 = nd.p_x[ox], qz0 = nd.q_z[oz];
      const int tcx0 = nd.tx[idx >= 0 ? idx : 0], tcz0 = nd.tz[idz >= 0 ? idz : 0];
      if (j != DD_NONE) {
        const int tc = idx >= 0 ? tcx0 : 0;
        if (tc != 1) {
          ++viol;
          const float qn = qx0 - eta * (tc - 1);
          nd.q_x[ox] = qn;
          if (j >= i + 3) {
            const float sv = w_x * (px0 - prm.th_s) - qn;
            if (nd.s_x) nd.s_x[fold_sidx(false, L1, Wx, i, j)] = sv;
            if (nd.s_xs) nd.s_xs[fold_sidx(true, L1, Wx, i, j)] = sv;
          }
        }
      }
      for (uint32_t e = pe0; e < pe1; ++e) {
        if (!nd.cx_flag[e]) continue;
        const uint32_t jj = nd.px_j[e];
        const int tc = nd.tx[e];
        if (j != jj && tc != 0) {
          ++viol;
          const size_t o = (size_t)i * L1 + jj;
          const float qn = nd.q_x[o] - eta * tc;
          nd.q_x[o] = qn;
          if (jj >= i + 3) {  // shorter spans stay 0 (dd_fill_scores)
            const float sv = w_x * (nd.p_x[o] - prm.th_s) - qn;
            if (nd.s_x) nd.s_x[fold_sidx(false, L1, Wx, i, jj)] = sv;
            if (nd.s_xs) nd.s_xs[fold_sidx(true, L1, Wx, i, jj)] = sv;
          }
        }
      }
      if (kz != DD_NONE) {
        const int tc = idz >= 0 ? tcz0 : 0;
        if (tc > 1) ++viol;
        const float v = qz0 - eta * (1 - tc);
        const float qn = (0.0f < v) ? v : 0.0f;
        nd.q_z[oz] = qn;
        nd.qz_s[nw_idx(L1, Wz, i + 1, kz + 1)] = qn;
      }
      for (uint32_t e = ce0; e < ce1; ++e) {
        const uint32_t kk = nd.cz_k[e];
        if (kz != kk) {
          const int tc = nd.tz[e];
          if (tc > 0) ++viol;
          const float v = nd.q_z[(size_t)i * L2 + kk] + eta * tc;
          const float qn = (0.0f < v) ? v : 0.0f;
          nd.q_z[(size_t)i * L2 + kk] = qn;
          nd.qz_s[nw_idx(L1, Wz, i + 1, kk + 1)] = qn;
        }
      }
    }
    for (uint32_t k = (tid + nt / 2) % nt; k < L2; k += nt) {  // the upper half of the workgroup starts on y while the lower half is on x
      const uint32_t l = nd.y[k];
      const uint32_t ye0 = nd.py_ptr[k], ye1 = nd.py_ptr[k + 1];
      const size_t oy = (size_t)k * L2 + (l != DD_NONE ? l : 0u);
      const int32_t idy = nd.ymap[oy];
      const float qy0 = nd.q_y[oy], py0 = nd.p_y[oy];
      const int tcy0 = nd.ty[idy >= 0 ? idy : 0];
      if (l != DD_NONE) {
        const int tc = idy >= 0 ? tcy0 : 0;
        if (tc != 1) {
          ++viol;
          const float qn = qy0 - eta * (tc - 1);
          nd.q_y[oy] = qn;
          if (l >= k + 3) {
            const float sv = w_y * (py0 - prm.th_s) - qn;
            if (nd.s_y) nd.s_y[fold_sidx(false, L2, Wy, k, l)] = sv;
            if (nd.s_ys) nd.s_ys[fold_sidx(true, L2, Wy, k, l)] = sv;
          }
        }
      }
      for (uint32_t e = ye0; e < ye1; ++e) {
        if (!nd.cy_flag[e]) continue;
        const uint32_t ll = nd.py_l[e];
        const int tc = nd.ty[e];
        if (l != ll && tc != 0) {
          ++viol;
          const size_t o = (size_t)k * L2 + ll;
          const float qn = nd.q_y[o] - eta * tc;
          nd.q_y[o] = qn;
          if (ll >= k + 3) {
            const float sv = w_y * (nd.p_y[o] - prm.th_s) - qn;
            if (nd.s_y) nd.s_y[fold_sidx(false, L2, Wy, k, ll)] = sv;
            if (nd.s_ys) nd.s_ys[fold_sidx(true, L2, Wy, k, ll)] = sv;
          }
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
      if (!stop && prm.budget && t + 1 != prm.t_max && wall_clock64() >= deadline) stop = 2;  // out of time: pause here
      s_stop = stop;
    }
    __syncthreads();
    DD_TICK(5);
    if (s_stop == 2) { paused = true; ++t; break; }  // to be continued, like a node whose slice ran out
    if (s_stop) break;
    if (prm.slice && ++ran == prm.slice && t + 1 != prm.t_max) { paused = true; ++t; break; }  // to be continued
  }
  if (tid == 0) {
    if (split) sync_store(&nd.sync[0], DD_SYNC_EXIT);  // the folders leave; a resumed launch starts them again
    nd.info[6] = 1;
    nd.info[7] = paused ? 1u : 0u;
    nd.info[1] = t;  // iterations done; while paused, the next iteration
    if (paused) {
      nd.fstate[0] = c; nd.fstate[1] = eta; nd.fstate[2] = s_prev;
    } else {
      *nd.score = s_prev;
      nd.info[2] = violated;
      nd.info[3] = s_bad ? 1u : 0u;
    }
    if (prm.stamps)
    {
      for (int k = 0; k < 6; ++k) nd.info[8 + k] = (resume ? nd.info[8 + k] : 0u) + (uint32_t)tk[k];
      nd.info[14] = (resume ? nd.info[14] : 0u) + (uint32_t)s_tky;
      nd.info[15] = (resume ? nd.info[15] : 0u) + (uint32_t)s_tkz;
    }
    // one word per node of the launch: a single copy tells the host who is done; 2 = parked because its folders were lost
    if (paused_out) paused_out[blockIdx.x] = paused ? (s_lost ? 2u : 1u) : 0u;
  }
}

// ------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------
// dynamic-LDS opt-in above 64 KB, once per device and kernel
static int lds_optin(const void* fn, int slot, size_t bytes, size_t budget = kDdLdsBudget) {
  static bool done[5][16] = {{false}};
  int dev = 0;
  if (hip_check(hipGetDevice(&dev))) return DAFS_HIP_ENODEV;
  if (bytes <= 64 * 1024) return DAFS_HIP_OK;
  if (dev < 0 || dev >= 16 || !done[slot][dev]) {
    if (hip_check(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)budget))) return DAFS_HIP_ELAUNCH;
    if (dev >= 0 && dev < 16) done[slot][dev] = true;
  }
  return DAFS_HIP_OK;
}

int dd_avg_launch(const dd_node* d_nodes, uint32_t nnodes, uint32_t max_len, mp_store_dev mp, bp_store_dev bp, int one_row, int coop, hipStream_t st) {
  if (!nnodes) return DAFS_HIP_OK;
  // one accumulator row per wavefront in LDS: four wavefronts per workgroup while four rows fit, then two, then one
  const size_t budget = 124 * 1024;  // the kernel has 32 KB of static staging areas
  const uint32_t row_cap = (max_len + 63) & ~63u;
  uint32_t rows = one_row ? 1 : 4;
  while (rows > 1 && (size_t)rows * row_cap * sizeof(float) > budget) rows >>= 1;
  const size_t lds = (size_t)rows * row_cap * sizeof(float);
  if (lds > budget) return DAFS_HIP_ETOOLONG;  // beyond ~31 000 columns
  int rc = lds_optin((const void*)k_node_avg, 0, lds, budget);
  if (rc) return rc;
  // coop: four wavefronts per workgroup (the p_x / p_y rows keep a wavefront each, four rows per workgroup; a p_z row
  // takes the whole workgroup, so the grid has a workgroup per row of the longest alignment)
  if (coop && rows == 4) {
    STAGE_LAUNCH(ST_NODE_AVG, st) hipLaunchKernelGGL(k_node_avg, dim3(max_len, nnodes, 3), dim3(256), lds, st, d_nodes, mp, bp, row_cap, 1u);
  } else {
    STAGE_LAUNCH(ST_NODE_AVG, st) hipLaunchKernelGGL(k_node_avg, dim3((max_len + rows - 1) / rows, nnodes, 3), dim3(64 * rows), lds, st, d_nodes, mp, bp, row_cap, 0u);
  }
  return hip_check(hipGetLastError()) ? DAFS_HIP_ELAUNCH : DAFS_HIP_OK;
}
int dd_lists_launch(const dd_node* d_nodes, uint32_t nnodes, uint32_t max_len1, dd_params prm, uint32_t* d_ncbp, hipStream_t st) {
  if (!nnodes) return DAFS_HIP_OK;
  // wide alignments: the dense scans on grids over the rows (DAFS_HIP_DD_LISTS_WIDE=n: from n columns on; tests force it)
  uint32_t wide_from = 1536;
  if (const char* e = getenv("DAFS_HIP_DD_LISTS_WIDE")) wide_from = (uint32_t)atoi(e);
  if (max_len1 && max_len1 >= wide_from) {
    const dim3 rows((max_len1 + 7) / 8, nnodes, 3);
    STAGE_LAUNCH(ST_NODE_LISTS, st) hipLaunchKernelGGL(k_lists_rows, rows, dim3(512), 0, st, d_nodes, prm.th_a, 0);
    if (hip_check(hipGetLastError())) return DAFS_HIP_ELAUNCH;
    STAGE_LAUNCH(ST_NODE_LISTS, st) hipLaunchKernelGGL(k_lists_scan, dim3(nnodes, 3), dim3(DD_THREADS), 0, st, d_nodes);
    if (hip_check(hipGetLastError())) return DAFS_HIP_ELAUNCH;
    STAGE_LAUNCH(ST_NODE_LISTS, st) hipLaunchKernelGGL(k_lists_rows, rows, dim3(512), 0, st, d_nodes, prm.th_a, 1);
    if (hip_check(hipGetLastError())) return DAFS_HIP_ELAUNCH;
    STAGE_LAUNCH(ST_NODE_LISTS, st) hipLaunchKernelGGL(k_node_lists_tail, dim3(nnodes), dim3(DD_THREADS), 0, st, d_nodes, prm, d_ncbp);
    return hip_check(hipGetLastError()) ? DAFS_HIP_ELAUNCH : DAFS_HIP_OK;
  }
  const size_t budget = 136 * 1024;  // 18 KB of static scan areas
  size_t lds = max_len1 ? ((size_t)max_len1 + 4) * 4 : 0;  // 0: forced to the HBM search (tests)
  if (lds > budget) lds = budget;  // longer row pointer arrays are searched in HBM
  int rc = lds_optin((const void*)k_node_lists, 1, lds, budget);
  if (rc) return rc;
  // four workgroups per node when the launch is small (the nodes of the tree's critical chain are opened one or two at a time)
  const uint32_t parts = (nnodes <= 16 && !getenv("DAFS_HIP_DD_LISTS1")) ? 4u : 1u;  // the env switch is a tuning aid
  STAGE_LAUNCH(ST_NODE_LISTS, st) hipLaunchKernelGGL(k_node_lists, dim3(nnodes, parts), dim3(DD_THREADS), lds, st, d_nodes, prm, d_ncbp, (uint32_t)lds, parts);
  return hip_check(hipGetLastError()) ? DAFS_HIP_ELAUNCH : DAFS_HIP_OK;
}
int dd_cbp_fill_launch(const dd_node* d_nodes, uint32_t nnodes, uint32_t max_len1, dd_params prm, hipStream_t st) {
  if (!nnodes) return DAFS_HIP_OK;
  const size_t budget = 150 * 1024;
  size_t lds = max_len1 ? ((size_t)max_len1 + 4) * 4 : 0;
  if (lds > budget) lds = budget;
  int rc = lds_optin((const void*)k_node_cbp_fill, 2, lds, budget);
  if (rc) return rc;
  STAGE_LAUNCH(ST_NODE_CBP_FILL, st) hipLaunchKernelGGL(k_node_cbp_fill, dim3(nnodes), dim3(DD_THREADS), lds, st, d_nodes, prm, (uint32_t)lds);
  return hip_check(hipGetLastError()) ? DAFS_HIP_ELAUNCH : DAFS_HIP_OK;
}
// The result words of every node of a launch (x, y, z, score, info: carved back to back in the node's block) gathered
// into one buffer, so that one copy brings the results of all the nodes that finished in the launch.
__global__ __launch_bounds__(256) void k_node_pack(const dd_node* nodes, const uint32_t* off_words, uint32_t* out) {
  const dd_node& nd = nodes[blockIdx.x];
  const uint32_t* src = nd.x;
  const uint32_t n = (uint32_t)((nd.info + 16) - nd.x);
  uint32_t* dst = out + off_words[blockIdx.x];
  for (uint32_t k = threadIdx.x; k < n; k += blockDim.x) dst[k] = src[k];
}
int dd_pack_launch(const dd_node* d_nodes, uint32_t nnodes, const uint32_t* d_off, uint32_t* d_out, hipStream_t st) {
  if (!nnodes) return DAFS_HIP_OK;
  STAGE_LAUNCH(ST_NODE_PACK, st) hipLaunchKernelGGL(k_node_pack, dim3(nnodes), dim3(256), 0, st, d_nodes, d_off, d_out);
  return hip_check(hipGetLastError()) ? DAFS_HIP_ELAUNCH : DAFS_HIP_OK;
}

int dd_solve_launch(const dd_node* d_nodes, uint32_t nnodes, dd_params prm, size_t lds_bytes, bool split, uint32_t* d_paused, hipStream_t st) {
  if (!nnodes) return DAFS_HIP_OK;
  if (lds_bytes > kDdLdsBudget) return DAFS_HIP_EINVAL;
  {
    const int rc = lds_optin((const void*)k_dd_solve, 3, lds_bytes);
    if (rc) return rc;
  }
  // split mode needs the three workgroups of a node on the machine together: the caller keeps 3 * nnodes within the CU count
  STAGE_LAUNCH(ST_DD_SOLVE, st) hipLaunchKernelGGL(k_dd_solve, dim3(nnodes, split ? 3 : 1), dim3(DD_SOLVE_THREADS), lds_bytes, st, d_nodes, prm, d_paused);
  return hip_check(hipGetLastError()) ? DAFS_HIP_ELAUNCH : DAFS_HIP_OK;
}
int nussinov_launch(uint32_t L, const float* p, const float* q, float w, float th, nuss_ws ws, uint32_t* ss, float* score, hipStream_t st) {
  uint32_t heads = DD_NONE;
  size_t lds = 0;
  if (!getenv("DAFS_HIP_NUSS_GLOBAL"))  // tests: keep the span-ordered form on global tables
    for (uint32_t K : {4u, 2u, 0u})
      if ((size_t)dd_wg_words(L, K) * 4 + 16 <= kDdLdsBudget) { heads = K; lds = (size_t)dd_wg_words(L, K) * 4 + 16; break; }
  if (lds > 64 * 1024) {
    const int rc = lds_optin((const void*)k_nussinov_single, 4, lds);
    if (rc) return rc;
  }
  STAGE_LAUNCH(ST_NUSSINOV_SINGLE, st) hipLaunchKernelGGL(k_nussinov_single, dim3(1), dim3(DD_THREADS), lds, st, L, p, q, w, th, ws, ss, score, heads);
  return hip_check(hipGetLastError()) ? DAFS_HIP_ELAUNCH : DAFS_HIP_OK;
}
int nussinov_dense_launch(uint32_t L, const float* p, const float* q, float w, float th, float* dp, uint32_t* tr, uint32_t* stack, uint32_t* ss,
                          float* score, hipStream_t st) {
  hipLaunchKernelGGL(k_nussinov_dense, dim3(1), dim3(DD_THREADS), 0, st, L, p, q, w, th, dp, tr, stack, ss, score);
  return hip_check(hipGetLastError()) ? DAFS_HIP_ELAUNCH : DAFS_HIP_OK;
}
int nw_launch(uint32_t L1, uint32_t L2, const float* p, const float* q, float th, uint32_t* env, int compute_env,
              float* dp, uint8_t* tr, uint32_t* al, float* score, hipStream_t st) {
  STAGE_LAUNCH(ST_NW_SINGLE, st) hipLaunchKernelGGL(k_nw_single, dim3(1), dim3(DD_THREADS), 0, st, L1, L2, p, q, th, env, compute_env, dp, tr, al, score);
  return hip_check(hipGetLastError()) ? DAFS_HIP_ELAUNCH : DAFS_HIP_OK;
}

}  // namespace dafs
