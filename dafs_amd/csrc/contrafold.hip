// dafs_amd/csrc/contrafold.hip -- CONTRAfold base-pairing posteriors on the GPU.
//
// Replaces CONTRAFOLD::InferenceEngine<float>::{ComputeInside, ComputeOutside, ComputePosterior}
// (reference src/contrafold/InferenceEngine.ipp:3356-3722, 3731-4080, 4498-4821; live feature set of
// src/contrafold/Config.hpp:156-179: FC / FM / FM1 / F5 grammar) and the CONTRAfold::calculate
// adapter (src/fold.cpp:174-207).
//
// One workgroup folds one sequence; all tables stay in HBM/L2 (7 triangular float tables).
//   inside  : cells of equal span j-i are independent -> one span per barrier, a lane per cell;
//             F5 is a chain over j and is folded by one lane from terms the block prepares.
//   outside : the reference SCATTERS into FCo/FMo/FM1o while sweeping i up, j down.  Each target
//             here GATHERS its addends in exactly the order that sweep would have delivered them
//             (derivation in DESIGN.md), so the log-sum-exp chains round identically; spans now
//             run from long to short, again one per barrier.
//   posterior: every pair gathers its Fast_Exp terms in the reference's visiting order; fully
//             parallel over pairs.
// Log-sum-exp is the reference's 8-piece cubic (contra_math.h); -ffp-contract=off.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/dafs_hip.h"
#include "contra_math.h"
#include "contrafold.h"
#include "hip_util.h"

namespace dafs {

#define CF_MAX_SINGLE 30  // C_MAX_SINGLE_LENGTH, Config.hpp:213
#define CF_THREADS 256

struct cf_ctx {  // per-workgroup view
  int L;
  const int* s;          // symbol 0..4 per position 0..L+1 (s[0] = s[L+1] = 4)
  const int* map;        // constraint mapping per position (-1 unknown, 0 unpaired, else partner)
  const int* cum;        // cum[t] = number of positions 1..t that may NOT be unpaired
  const int* off;        // row offsets of the triangular tables
  const cf_params* P;    // score tables (LDS copy)
};

__device__ __forceinline__ bool cf_comp(int a, int b) {  // AU, GU, CG (InferenceEngine ctor)
  return (a == 0 && b == 3) || (a == 3 && b == 0) || (a == 2 && b == 3) || (a == 3 && b == 2) || (a == 1 && b == 2) || (a == 2 && b == 1);
}
// allow_paired[offset[i]+j] after LoadSequence (:947-1097) and UseConstraints (:1870-1902)
__device__ __forceinline__ bool cf_allow_paired(const cf_ctx& c, int i, int j) {
  if (i <= 0 || j <= i || j > c.L) return false;
  const int mi = c.map[i], mj = c.map[j];
  return (mi == -1 || mi == j) && (mj == -1 || mj == i) && cf_comp(c.s[i], c.s[j]);
}
// every position in (lo, hi] may be unpaired
__device__ __forceinline__ bool cf_all_unpaired(const cf_ctx& c, int lo, int hi) { return c.cum[hi] - c.cum[lo] == 0; }
__device__ __forceinline__ bool cf_unpaired_pos(const cf_ctx& c, int t) { return c.cum[t] - c.cum[t - 1] == 0; }

#define S_(i) (c.s[i])
__device__ __forceinline__ float cf_junction_a(const cf_ctx& c, int i, int j) {  // :1927-1956
  return 0.0f + c.P->helix_closing[S_(i) * 5 + S_(j + 1)] +
         (i < c.L ? c.P->dangle_left[(S_(i) * 5 + S_(j + 1)) * 5 + S_(i + 1)] : 0.0f) +
         (j > 0 ? c.P->dangle_right[(S_(i) * 5 + S_(j + 1)) * 5 + S_(j)] : 0.0f);
}
__device__ __forceinline__ float cf_junction_b(const cf_ctx& c, int i, int j) {  // :2004-2030
  return 0.0f + c.P->helix_closing[S_(i) * 5 + S_(j + 1)] + c.P->terminal_mismatch[((S_(i) * 5 + S_(j + 1)) * 5 + S_(i + 1)) * 5 + S_(j)];
}
__device__ __forceinline__ float cf_base_pair(const cf_ctx& c, int i, int j) { return 0.0f + c.P->base_pair[S_(i) * 5 + S_(j)]; }  // :2060-2084
__device__ __forceinline__ float cf_helix_stacking(const cf_ctx& c, int i, int j) {  // :217-230
  return c.P->helix_stacking[((S_(i) * 5 + S_(j)) * 5 + S_(i + 1)) * 5 + S_(j - 1)];
}
__device__ __forceinline__ float cf_hairpin(const cf_ctx& c, int i, int j) {  // :2123-2153
  return 0.0f + cf_junction_b(c, i, j) + c.P->cache_hairpin[min(j - i, 30)];
}
__device__ __forceinline__ float cf_single_nuc(const cf_ctx& c, int i, int j, int p, int q) {  // :2290-2361
  const int l1 = p - i, l2 = j - q;
  return 0.0f + 0.0f + (l1 == 0 && l2 == 1 ? c.P->bulge_0x1[S_(j)] : 0.0f) + (l1 == 1 && l2 == 0 ? c.P->bulge_1x0[S_(i + 1)] : 0.0f) +
         (l1 == 1 && l2 == 1 ? c.P->internal_1x1[S_(i + 1) * 5 + S_(j)] : 0.0f);
}
#define MULTI_UNPAIRED (c.P->multi_unpaired + 0.0f)
#define EXT_UNPAIRED (c.P->external_unpaired + 0.0f)

// ---------------------------------------------------------------------------------------------
// inside cell (i,j), InferenceEngine.ipp:3392-3688
// ---------------------------------------------------------------------------------------------
__device__ void cf_inside_cell(const cf_ctx& c, int i, int j, float* FCi, float* FMi, float* FM1i) {
  const int L = c.L;
  const int* off = c.off;
  float FM2i = CONTRA_NEG_INF;
  if (i + 2 <= j)
    for (int k = i + 1; k < j; k++) FM2i = contra_lpe(FM2i, FM1i[off[i] + k] + FMi[off[k] + j]);
  if (0 < i && j < L && cf_allow_paired(c, i, j + 1)) {
    float sum = CONTRA_NEG_INF;
    if (cf_all_unpaired(c, i, j)) sum = contra_lpe(sum, cf_hairpin(c, i, j));
    const float score_helix = (i + 2 <= j ? cf_base_pair(c, i + 1, j) + cf_helix_stacking(c, i, j + 1) : 0.0f);
    const float score_other = cf_junction_b(c, i, j);
    const int pmax = min(i + CF_MAX_SINGLE, j);
    for (int p = i; p <= pmax; p++) {
      if (p > i && !cf_unpaired_pos(c, p)) break;
      const int q_min = max(p + 2, p - i + j - CF_MAX_SINGLE);
      const float* FCptr = FCi + off[p + 1] - 1;
      for (int q = j; q >= q_min; q--) {
        if (q < j && !cf_unpaired_pos(c, q + 1)) break;
        if (!cf_allow_paired(c, p + 1, q)) continue;
        const float score = (p == i && q == j)
                                ? (score_helix + FCptr[q])
                                : (score_other + c.P->cache_single[(p - i) * 31 + (j - q)] + FCptr[q] + cf_base_pair(c, p + 1, q) +
                                   cf_junction_b(c, q, p) + cf_single_nuc(c, i, j, p, q));
        sum = contra_lpe(sum, score);
      }
    }
    sum = contra_lpe(sum, FM2i + cf_junction_a(c, i, j) + c.P->multi_paired + c.P->multi_base);
    FCi[off[i] + j] = sum;
  }
  if (0 < i && i + 2 <= j && j < L) {
    float sum = CONTRA_NEG_INF;
    if (cf_allow_paired(c, i + 1, j))
      sum = contra_lpe(sum, FCi[off[i + 1] + j - 1] + cf_junction_a(c, j, i) + c.P->multi_paired + cf_base_pair(c, i + 1, j));
    if (cf_unpaired_pos(c, i + 1)) sum = contra_lpe(sum, FM1i[off[i + 1] + j] + MULTI_UNPAIRED);
    FM1i[off[i] + j] = sum;
    float sm = CONTRA_NEG_INF;
    sm = contra_lpe(sm, FM2i);
    if (cf_unpaired_pos(c, j)) sm = contra_lpe(sm, FMi[off[i] + j - 1] + MULTI_UNPAIRED);
    sm = contra_lpe(sm, sum);
    FMi[off[i] + j] = sm;
  }
}

// ---------------------------------------------------------------------------------------------
// outside cell (a,b): gathers, in the reference's delivery order, everything ComputeOutside
// (:3731-4080) adds to FMo[a][b], FCo[a][b], FM1o[a][b]; then forms FM2o(a,b)
// ---------------------------------------------------------------------------------------------
__device__ void cf_outside_cell(const cf_ctx& c, int a, int b, const float* FCi, const float* FMi, const float* FM1i, const float* F5i,
                                const float* F5o, float* FCo, float* FMo, float* FM1o, float* FM2o) {
  const int L = c.L;
  const int* off = c.off;
  // ---- FMo[a][b]: block 4 of sources (i,b), i = 0..a-1 (:4040-4068), then block 1 of source (a,b+1) (:3770-3777)
  float fmo = CONTRA_NEG_INF;
  if (a < b)
    for (int i = 0; i < a; i++) fmo = contra_lpe(fmo, FM2o[off[i] + b] + FM1i[off[i] + a]);
  if (0 < a && a + 2 <= b + 1 && b + 1 < L && cf_unpaired_pos(c, b + 1)) fmo = contra_lpe(fmo, FMo[off[a] + b + 1] + MULTI_UNPAIRED);
  FMo[off[a] + b] = fmo;

  // ---- FCo[a][b] (only cells whose closing pair (a, b+1) is allowed ever receive anything)
  float fco = CONTRA_NEG_INF;
  const bool pair_ok = (0 < a && b < L && cf_allow_paired(c, a, b + 1));
  if (pair_ok) {
    const int p = a - 1, q = b + 1;
    {  // exterior loop, first sweep (:3754-3766): k = p, j = q
      const float temp = F5o[q] + c.P->external_paired + cf_base_pair(c, p + 1, q) + cf_junction_a(c, q, p);
      fco = contra_lpe(fco, temp + F5i[p]);
    }
    for (int i = max(1, p - CF_MAX_SINGLE); i <= p; i++) {
      const int l1 = p - i;
      if (l1 > 0 && !cf_all_unpaired(c, i, p)) continue;
      const int jmax = min(L - 1, q + CF_MAX_SINGLE - l1);
      for (int j = jmax; j >= q; j--) {
        const int l2 = j - q;
        if (i == p && j == q) {
          // source (p,q): block 2 (:3787-3789) comes before its own single-branch scatter
          if (0 < p && p + 2 <= q && q < L)
            fco = contra_lpe(fco, FM1o[off[p] + q] + cf_junction_a(c, q, p) + c.P->multi_paired + cf_base_pair(c, p + 1, q));
          if (cf_allow_paired(c, i, j + 1)) {
            const float score_helix = FCo[off[i] + j] + cf_base_pair(c, i + 1, j) + cf_helix_stacking(c, i, j + 1);
            fco = contra_lpe(fco, score_helix);
          }
          continue;
        }
        if (!cf_allow_paired(c, i, j + 1)) continue;
        if (l2 > 0 && !cf_all_unpaired(c, q, j)) continue;
        const float score_other = FCo[off[i] + j] + cf_junction_b(c, i, j);
        fco = contra_lpe(fco, score_other + c.P->cache_single[l1 * 31 + l2] + cf_base_pair(c, p + 1, q) + cf_junction_b(c, q, p) +
                                  cf_single_nuc(c, i, j, p, q));
      }
    }
    FCo[off[a] + b] = fco;
  }

  // ---- FM1o[a][b]: block 2 of source (a-1,b), block 4 of sources (a,j) j = L..b+1, block 1 of source (a,b)
  float fm1o = CONTRA_NEG_INF;
  if (0 < a - 1 && a - 1 + 2 <= b && b < L && cf_unpaired_pos(c, a)) fm1o = contra_lpe(fm1o, FM1o[off[a - 1] + b] + MULTI_UNPAIRED);
  if (a < b)
    for (int j = L; j > b; j--) fm1o = contra_lpe(fm1o, FM2o[off[a] + j] + FMi[off[b] + j]);
  const bool live = (0 < a && a + 2 <= b && b < L);
  if (live) fm1o = contra_lpe(fm1o, fmo);
  FM1o[off[a] + b] = fm1o;

  // ---- FM2o(a,b), the value the reference holds locally while scattering from (a,b)
  float fm2o = CONTRA_NEG_INF;
  if (live) fm2o = contra_lpe(fm2o, fmo);
  if (pair_ok) fm2o = contra_lpe(fm2o, fco + cf_junction_a(c, a, b) + c.P->multi_paired + c.P->multi_base);
  FM2o[off[a] + b] = fm2o;
}

// ---------------------------------------------------------------------------------------------
// posterior of pair (a, q) (index off[a]+q), ComputePosterior :4498-4821
// ---------------------------------------------------------------------------------------------
__device__ float cf_posterior_cell(const cf_ctx& c, int a, int q, const float* FCi, const float* F5i, const float* F5o, const float* FCo,
                                   const float* FM1o, float Z) {
  if (!cf_allow_paired(c, a, q)) return 0.0f;
  const int L = c.L, p = a - 1;
  const int* off = c.off;
  float acc = 0.0f;
  const float inner = FCi[off[p + 1] + q - 1];  // FCptr[q]
  if (q < L) {  // single-branch sources need j >= q and j < L
    for (int i = p; i >= max(1, p - CF_MAX_SINGLE); i--) {
      const int l1 = p - i;
      if (l1 > 0 && !cf_all_unpaired(c, i, p)) break;  // larger l1 only adds more positions
      const int jmax = min(L - 1, q + CF_MAX_SINGLE - l1);
      for (int j = q; j <= jmax; j++) {
        const int l2 = j - q;
        if (l2 > 0 && !cf_all_unpaired(c, q, j)) break;
        if (cf_allow_paired(c, i, j + 1)) {
          const float outside = FCo[off[i] + j] - Z;
          float term;
          if (i == p && j == q) term = contra_exp((outside + cf_base_pair(c, i + 1, j) + cf_helix_stacking(c, i, j + 1)) + inner);
          else
            term = contra_exp((outside + cf_junction_b(c, i, j)) + c.P->cache_single[l1 * 31 + l2] + inner + cf_base_pair(c, p + 1, q) +
                              cf_junction_b(c, q, p) + cf_single_nuc(c, i, j, p, q));
          acc += term;
        }
        if (i == p && j == q && 0 < p && p + 2 <= q)  // multi-loop closing pair, :4741-4745 (source (p,q), after its single-branch block)
          acc += contra_exp(FM1o[off[p] + q] + inner + cf_junction_a(c, q, p) + c.P->multi_paired + cf_base_pair(c, p + 1, q) - Z);
      }
    }
  }
  // exterior, :4750-4760
  acc += contra_exp((F5o[q] - Z) + F5i[p] + inner + c.P->external_paired + cf_base_pair(c, p + 1, q) + cf_junction_a(c, q, p));
  const float m = acc < 0.0f ? 0.0f : acc;  // Clip, Utilities.ipp:136
  return (1.0f < m) ? 1.0f : m;
}

// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(CF_THREADS) void k_contrafold(cf_batch B) {
  __shared__ cf_params sP;
  __shared__ float s_terms[CF_THREADS];
  {
    const float* src = (const float*)B.params;
    float* dst = (float*)&sP;
    for (uint32_t k = threadIdx.x; k < sizeof(cf_params) / 4; k += blockDim.x) dst[k] = src[k];
  }
  const uint32_t x = blockIdx.x;
  const cf_seq sq = B.seqs[x];
  const int L = (int)sq.len;
  const int tid = threadIdx.x, nt = blockDim.x;
  // symbols, constraint map, prefix counts and row offsets are read in every inner-loop test:
  // keep them in LDS (4*(L+2) ints)
  extern __shared__ int s_ints[];
  int* s = s_ints;               // L+2
  int* map = s + (L + 2);        // L+2
  int* cum = map + (L + 2);      // L+2
  int* off = cum + (L + 2);      // L+2
  float* F = B.fws + sq.fws_off;
  const size_t SZ = (size_t)(L + 1) * (L + 2) / 2;
  float *FCi = F, *FMi = F + SZ, *FM1i = F + 2 * SZ, *FCo = F + 3 * SZ, *FMo = F + 4 * SZ, *FM1o = F + 5 * SZ, *FM2o = F + 6 * SZ;
  float *F5i = F + 7 * SZ, *F5o = F5i + (L + 1);
  float* post = B.post + sq.post_off;

  // LoadSequence (:947-1097): symbols, row offsets, constraint bookkeeping
  for (int i = tid; i <= L + 1; i += nt) {
    int sym = 4;
    if (i >= 1 && i <= L) {
      const uint8_t ch = B.codes[sq.code_off + i - 1];  // ProbCons class code: A C G U T N other
      sym = ch < 4 ? (int)ch : 4;                       // CONTRAfold alphabet is "ACGU" only (T is not U here)
    }
    s[i] = sym;
    map[i] = (i >= 1 && i <= L && sq.has_constraint) ? B.cons[sq.cons_off + i] : -1;
    if (i <= L) off[i] = i * (2 * (L + 1) - i - 1) / 2;
  }
  for (size_t k = tid; k < 7 * SZ + 2 * (size_t)(L + 1); k += nt) F[k] = CONTRA_NEG_INF;
  __syncthreads();
  if (tid == 0) {
    int run = 0;
    cum[0] = 0;
    for (int i = 1; i <= L; ++i) { run += (map[i] == -1 || map[i] == 0) ? 0 : 1; cum[i] = run; }
    cum[L + 1] = run;
  }
  __syncthreads();
  cf_ctx c;
  c.L = L; c.s = s; c.map = map; c.cum = cum; c.off = off; c.P = &sP;

  // ---- inside: span ascending
  for (int d = 0; d <= L; ++d) {
    for (int i = tid; i + d <= L; i += nt) cf_inside_cell(c, i, i + d, FCi, FMi, FM1i);
    __syncthreads();
  }
  // F5i (:3692-3717): a chain over j; the block prepares the addends, one lane folds them in order
  if (tid == 0) F5i[0] = 0.0f;
  __syncthreads();
  for (int j = 1; j <= L; ++j) {
    for (int k0 = 0; k0 < j; k0 += nt) {
      const int k = k0 + tid;
      float term = CONTRA_NEG_INF;
      bool on = false;
      if (k < j && cf_allow_paired(c, k + 1, j)) {
        on = true;
        term = F5i[k] + FCi[off[k + 1] + j - 1] + sP.external_paired + cf_base_pair(c, k + 1, j) + cf_junction_a(c, j, k);
      }
      s_terms[tid] = on ? term : __builtin_nanf("");  // NaN marks "no addend"
      __syncthreads();
      if (tid == 0) {
        float sum = (k0 == 0) ? CONTRA_NEG_INF : F5i[j];
        if (k0 == 0 && cf_unpaired_pos(c, j)) sum = contra_lpe(sum, F5i[j - 1] + EXT_UNPAIRED);
        const int n = min(nt, j - k0);
        for (int u = 0; u < n; ++u) {
          const float v = s_terms[u];
          if (v == v) sum = contra_lpe(sum, v);
        }
        F5i[j] = sum;
      }
      __syncthreads();
    }
  }
  const float Z = F5i[L];

  // ---- outside.  F5o first (:3746-3767): lane k owns F5o[k]; at step tau all lanes take the
  // addend of j = L - tau + 1, whose F5o[j] was completed in the previous steps.
  if (tid == 0) F5o[L] = 0.0f;
  __syncthreads();
  for (int j = L; j >= 1; --j) {
    const float f5oj = F5o[j];
    for (int k = tid; k < j; k += nt) {
      float v = F5o[k];
      if (k == j - 1 && cf_unpaired_pos(c, j)) v = contra_lpe(v, f5oj + EXT_UNPAIRED);
      if (cf_allow_paired(c, k + 1, j)) {
        const float temp = f5oj + sP.external_paired + cf_base_pair(c, k + 1, j) + cf_junction_a(c, j, k);
        v = contra_lpe(v, temp + FCi[off[k + 1] + j - 1]);
      }
      F5o[k] = v;
    }
    __syncthreads();
  }
  // main sweep: span descending
  for (int d = L; d >= 0; --d) {
    for (int a = tid; a + d <= L; a += nt) cf_outside_cell(c, a, a + d, FCi, FMi, FM1i, F5i, F5o, FCo, FMo, FM1o, FM2o);
    __syncthreads();
  }
  // ---- posterior, written in the reference's triangular layout (row i = 0..L, col j = i..L)
  for (size_t k = tid; k < SZ; k += nt) post[k] = 0.0f;
  __syncthreads();
  for (int a = 1; a <= L; ++a)
    for (int q = a + 1 + tid; q <= L; q += nt) post[off[a] + q] = cf_posterior_cell(c, a, q, FCi, F5i, F5o, FCo, FM1o, Z);
  if (tid == 0 && B.logz) B.logz[x] = Z;
}

// dense triangular posterior -> BP rows (i-1) -> (j-1, p) with p > th (fold.cpp:181-188)
__global__ __launch_bounds__(256) void k_bp_compact(cf_batch B, float th, const uint64_t* rp_off, uint32_t* out_rowptr, uint32_t* out_col,
                                                    float* out_val, uint64_t* out_off, uint32_t* out_nnz, unsigned long long* pool_top,
                                                    uint64_t pool_cap, int* status) {
  const uint32_t x = blockIdx.x;
  const cf_seq sq = B.seqs[x];
  const int L = (int)sq.len;
  const float* post = B.post + sq.post_off;
  uint32_t* rowptr = out_rowptr + rp_off[x];
  __shared__ unsigned long long s_off;
  __shared__ int s_ok;
  auto offs = [L](int i) { return i * (2 * (L + 1) - i - 1) / 2; };
  for (int i = 1 + threadIdx.x; i <= L; i += blockDim.x) {
    uint32_t n = 0;
    for (int j = i; j <= L; ++j) n += post[offs(i) + j] > th ? 1 : 0;
    rowptr[i] = n;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    rowptr[0] = 0;
    for (int i = 0; i < L; ++i) rowptr[i + 1] += rowptr[i];
    const uint32_t n = rowptr[L];
    const unsigned long long o = atomicAdd(pool_top, (unsigned long long)n);
    s_off = o;
    s_ok = o + n <= pool_cap;
    if (!s_ok) atomicExch(status, DAFS_HIP_EOVERFLOW);
    out_off[x] = o;
    out_nnz[x] = n;
  }
  __syncthreads();
  if (!s_ok) return;
  for (int i = 1 + threadIdx.x; i <= L; i += blockDim.x) {
    unsigned long long pos = s_off + rowptr[i - 1];
    for (int j = i; j <= L; ++j) {
      const float v = post[offs(i) + j];
      if (v > th) { out_col[pos] = (uint32_t)(j - 1); out_val[pos] = v; ++pos; }
    }
  }
}

int contrafold_launch(const cf_batch& B, uint32_t nseq, uint32_t max_len, hipStream_t st) {
  if (!nseq) return DAFS_HIP_OK;
  const size_t lds = 4 * ((size_t)max_len + 2) * sizeof(int);
  if (lds > 48 * 1024) return DAFS_HIP_ETOOLONG;
  hipLaunchKernelGGL(k_contrafold, dim3(nseq), dim3(CF_THREADS), lds, st, B);
  return hip_check(hipGetLastError()) ? DAFS_HIP_ELAUNCH : DAFS_HIP_OK;
}

int bp_compact_launch(const cf_batch& B, uint32_t nseq, float th, const uint64_t* rp_off, uint32_t* out_rowptr, uint32_t* out_col,
                      float* out_val, uint64_t* out_off, uint32_t* out_nnz, unsigned long long* pool_top, uint64_t pool_cap, int* status,
                      hipStream_t st) {
  if (!nseq) return DAFS_HIP_OK;
  hipLaunchKernelGGL(k_bp_compact, dim3(nseq), dim3(256), 0, st, B, th, rp_off, out_rowptr, out_col, out_val, out_off, out_nnz, pool_top,
                     pool_cap, status);
  return hip_check(hipGetLastError()) ? DAFS_HIP_ELAUNCH : DAFS_HIP_OK;
}

}  // namespace dafs
