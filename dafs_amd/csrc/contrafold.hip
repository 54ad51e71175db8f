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
#include "stage.h"

namespace dafs {

// Fast_LogPlusEquals of this file picks its cubic through an LDS table (contra_lpe_t): fewer instructions than the
// select tree at the price of one more LDS round trip on the dependency chain -- the kernel is bound by the
// instruction stream of the few wavefronts a sequence occupies (65 -> 58 ms at N=128).  Same values either way.
__shared__ contra_tables g_cf_tab;
__device__ __forceinline__ float cf_lpe(float x, float y) { return contra_lpe_t1(&g_cf_tab, x, y); }
#define CF_TABLES_INIT() do { contra_tables_init(&g_cf_tab, threadIdx.x); } while (0)


#define CF_MAX_SINGLE 30  // C_MAX_SINGLE_LENGTH, Config.hpp:213
#define CF_THREADS 256
// k_contrafold: the cells of a span are dealt round-robin to 16 wavefronts (cell -> lane*16 + wavefront), so a
// span of 150 cells keeps four wavefronts per SIMD busy with ~10 lanes each instead of three wavefronts in all:
// the inner loops are chains of dependent LDS lookups, and it is the number of wavefronts that hides them
#define CF_FOLD_THREADS 1024

#define CF_LDS __attribute__((address_space(3)))  // the ring is read on the hottest path: keep its loads ds_*, not flat_*
struct cf_ctx {  // per-workgroup view
  int L;
  const int* s;          // symbol 0..4 per position 0..L+1 (s[0] = s[L+1] = 4)
  const int* map;        // constraint mapping per position (-1 unknown, 0 unpaired, else partner)
  const int* cum;        // cum[t] = number of positions 1..t that may NOT be unpaired
  const int* off;        // row offsets of the triangular tables
  // Pairing partners by symbol: plist[y*L ..] = ascending positions whose symbol pairs with symbol y
  // (AU, GU, CG), pcnt[y*(L+2) + j] = how many of them are <= j.  Lets the single-branch loops
  // visit only pairable (p+1, q) instead of testing every q.
  const int* plist;
  const int* pcnt;
  CF_LDS float* ring;    // LDS ring of the last 33 spans of FC (inside) / FCo (outside): ring[(span % 33)*(L+1) + row]
  bool has_ring;         // false: no room for it (sequences beyond ~540 nt), FC/FCo are read from HBM/L2.  Kept as a flag: the
                         // null of the LDS address space is not the generic null, a pointer test after the cast is a trap
  const cf_params* P;    // score tables (LDS copy)
  bool free;             // no constraint string: every map entry is -1 and cum is all zero, the lookups are skipped
  // Term pool of the current span (unconstrained sequences): the single-branch addends of every cell, evaluated by the
  // whole workgroup before the cells fold them (cf_inside_terms / cf_outside_terms).  tbase[row] = start of the row's
  // list in `pool`, or -1 when the pool was full (the cell then walks its partners itself); tcnt[row] = its length.
  // Two buffers when there is room (pool_bufs == 2): the wavefronts that own no cell evaluate the terms of the next
  // span into one while the cells of this span fold the other (a span's terms only read FC values two or more spans
  // away, so they do not wait for this span).  `cur` = the buffer the cells read.
  bool has_pool;
  CF_LDS float* pool;
  CF_LDS int* tbase;
  CF_LDS int* tcnt;
  CF_LDS int* ptop;
  int pool_cap;   // floats per buffer
  int cur;
};
__device__ __forceinline__ CF_LDS int* cf_tbase(const cf_ctx& c, int buf) { return c.tbase + buf * (c.L + 1); }
__device__ __forceinline__ CF_LDS int* cf_tcnt(const cf_ctx& c, int buf) { return c.tcnt + buf * (c.L + 1); }
__device__ __forceinline__ CF_LDS float* cf_pool(const cf_ctx& c, int buf) { return c.pool + (size_t)buf * c.pool_cap; }
#define CF_RING 33

__device__ __forceinline__ bool cf_comp(int a, int b) {  // AU, GU, CG (InferenceEngine ctor)
  return (a == 0 && b == 3) || (a == 3 && b == 0) || (a == 2 && b == 3) || (a == 3 && b == 2) || (a == 1 && b == 2) || (a == 2 && b == 1);
}
// allow_paired[offset[i]+j] after LoadSequence (:947-1097) and UseConstraints (:1870-1902)
__device__ __forceinline__ bool cf_allow_paired(const cf_ctx& c, int i, int j) {
  if (i <= 0 || j <= i || j > c.L) return false;
  if (c.free) return cf_comp(c.s[i], c.s[j]);
  const int mi = c.map[i], mj = c.map[j];
  return (mi == -1 || mi == j) && (mj == -1 || mj == i) && cf_comp(c.s[i], c.s[j]);
}
// every position in (lo, hi] may be unpaired
__device__ __forceinline__ bool cf_all_unpaired(const cf_ctx& c, int lo, int hi) { return c.free || c.cum[hi] - c.cum[lo] == 0; }
__device__ __forceinline__ bool cf_unpaired_pos(const cf_ctx& c, int t) { return c.free || c.cum[t] - c.cum[t - 1] == 0; }

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
  if (l1 + l2 > 2 || (l1 | l2) > 1) return 0.0f;  // every addend below is the literal 0.0f then: the sum is +0.0f
  return 0.0f + 0.0f + (l1 == 0 && l2 == 1 ? c.P->bulge_0x1[S_(j)] : 0.0f) + (l1 == 1 && l2 == 0 ? c.P->bulge_1x0[S_(i + 1)] : 0.0f) +
         (l1 == 1 && l2 == 1 ? c.P->internal_1x1[S_(i + 1) * 5 + S_(j)] : 0.0f);
}
#define MULTI_UNPAIRED (c.P->multi_unpaired + 0.0f)
#define EXT_UNPAIRED (c.P->external_unpaired + 0.0f)

// ---------------------------------------------------------------------------------------------
// Single-branch addends, evaluated ahead of the chains that consume them.
//
// A cell's FC / FCo value is a chain sum = sum (+) term_1 (+) term_2 ... of up to 496 single-branch terms, and
// Fast_LogPlusEquals is not associative: the chain is serial.  The terms are not: each is a handful of table lookups
// that depend on the partner walk, not on the running sum.  Evaluated inside the chain's loop (one lane per cell,
// round 1), every term cost four dependent LDS round trips and ~60 instructions of a wavefront that had ten lanes at
// work -- ~1000 cycles per term, 86 % of the inside pass.  Here the whole workgroup evaluates the terms of a span
// first: one lane per (cell, left offset l1), 32 lanes per cell, each lane writing the terms of its partners into the
// cell's list in LDS in the order the chain takes them (the lists are laid out by a 32-lane prefix sum of the
// partner counts, which are two table lookups, and placed in the pool by an atomic bump); after a barrier the cell's
// lane folds the list with the sixteen-deep prefetch of cf_fold, so the chain sees one log-sum-exp per term and
// nothing else.  Same terms, same order, same arithmetic: the values are bit-identical to the in-loop form, which
// stays for constrained sequences and for cells whose list did not fit the pool.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float cf_fc_load(const cf_ctx& c, const float* FC, int row, int col) {  // FC[row][col], recent spans from LDS
  if (c.has_ring) return c.ring[((uint32_t)(col - row) % (uint32_t)CF_RING) * (uint32_t)(c.L + 1) + (uint32_t)row];
  return FC[c.off[row] + col];
}
__device__ __forceinline__ int cf_half_prefix(int n, int hl, int* total) {  // inclusive prefix over the 32 lanes of a half-wave
  int incl = n;
#pragma unroll
  for (int o = 1; o < 32; o <<= 1) {
    const int up = __shfl_up(incl, o, 32);
    if (hl >= o) incl += up;
  }
  *total = __shfl(incl, 31, 32);
  return incl;
}
// reserve room for a cell's list: lane 0 of the half bumps the pool, everyone gets the start (-1: no room)
__device__ __forceinline__ int cf_pool_reserve(const cf_ctx& c, int buf, int row, int total, int hl, bool wanted) {
  int base = -1;
  if (hl == 0) {
    if (wanted) {
      if (total > 0) {
        const int b = atomicAdd((int*)(c.ptop + buf), total);
        if (b + total <= c.pool_cap) base = b;
      } else {
        base = 0;
      }
    }
    cf_tbase(c, buf)[row] = base;
    cf_tcnt(c, buf)[row] = total;
  }
  return __shfl(base, 0, 32);
}

// inside, span d: the terms of cf_inside_cell's single-branch loop (:3433-3528), p = i + l1 ascending, partners q descending
// (threads t0 .. t0+tn-1 of the workgroup take part: whole wavefronts)
__device__ void cf_inside_terms(const cf_ctx& c, int d, const float* FCi, int buf, int t0, int tn) {
  const int L = c.L;
  const int nitems = (L - d + 1) * 32;
  const int hl = threadIdx.x & 31;
  for (int item = (int)threadIdx.x - t0; item < nitems; item += tn) {
    const int i = item >> 5, l1 = hl, j = i + d, p = i + l1;
    const bool closing = (0 < i && j < L && cf_allow_paired(c, i, j + 1));
    int n = 0, ehi = -1, sy = 4;
    if (closing && l1 <= CF_MAX_SINGLE && p <= j) {
      sy = c.s[p + 1];
      if (sy != 4) {
        const int q_min = max(p + 2, l1 + j - CF_MAX_SINGLE);
        ehi = c.pcnt[sy * (L + 2) + j] - 1;                       // partners <= j
        n = max(0, ehi + 1 - c.pcnt[sy * (L + 2) + q_min - 1]);   // ... that are >= q_min
      }
    }
    int total;
    const int incl = cf_half_prefix(n, hl, &total);
    const int base = cf_pool_reserve(c, buf, i, total, hl, closing);
    if (base < 0 || n == 0) continue;
    CF_LDS float* out = cf_pool(c, buf) + base + incl - n;
    const float score_helix = (i + 2 <= j ? cf_base_pair(c, i + 1, j) + cf_helix_stacking(c, i, j + 1) : 0.0f);
    const float score_other = cf_junction_b(c, i, j);
    const int* pl = c.plist + sy * L;
    const int sp = c.s[p];
    for (int k = 0; k < n; ++k) {
      const int q = pl[ehi - k];
      const float inner = cf_fc_load(c, FCi, p + 1, q - 1);
      const int sq_ = c.s[q], sq1 = c.s[q + 1];
      const float bp = 0.0f + c.P->base_pair[sy * 5 + sq_];
      const float jb = 0.0f + c.P->helix_closing[sq_ * 5 + sy] + c.P->terminal_mismatch[((sq_ * 5 + sy) * 5 + sq1) * 5 + sp];
      out[k] = (p == i && q == j) ? (score_helix + inner)
                                  : (score_other + c.P->cache_single[(p - i) * 31 + (j - q)] + inner + bp + jb + cf_single_nuc(c, i, j, p, q));
    }
  }
}

// outside, span d: the single-branch sources of target (a, b = a + d) in cf_outside_cell's order, i = p-30 .. p ascending
// (p = a - 1), partners j+1 descending
__device__ void cf_outside_terms(const cf_ctx& c, int d, const float* FCo, int buf, int t0, int tn) {
  const int L = c.L;
  const int nitems = (L - d + 1) * 32;
  const int hl = threadIdx.x & 31;
  for (int item = (int)threadIdx.x - t0; item < nitems; item += tn) {
    const int a = item >> 5, b = a + d, p = a - 1, q = b + 1;
    const int i = p - CF_MAX_SINGLE + hl, l1 = p - i;
    const bool pair_ok = (0 < a && b < L && cf_allow_paired(c, a, b + 1));
    int n = 0, ehi = -1, sy = 4;
    if (pair_ok && hl <= CF_MAX_SINGLE && i >= 1) {
      sy = c.s[i];
      if (sy != 4) {
        const int jmax = min(L - 1, q + CF_MAX_SINGLE - l1);
        const int lowest = (i == p) ? q + 2 : q + 1;
        ehi = c.pcnt[sy * (L + 2) + jmax + 1] - 1;
        n = max(0, ehi + 1 - c.pcnt[sy * (L + 2) + lowest - 1]);
      }
    }
    int total;
    const int incl = cf_half_prefix(n, hl, &total);
    const int base = cf_pool_reserve(c, buf, a, total, hl, pair_ok);
    if (base < 0 || n == 0) continue;
    CF_LDS float* out = cf_pool(c, buf) + base + incl - n;
    const float bp_pq = cf_base_pair(c, p + 1, q), jb_qp = cf_junction_b(c, q, p);
    const int* pl = c.plist + sy * L;
    const int si1 = c.s[i + 1];
    for (int k = 0; k < n; ++k) {
      const int j1 = pl[ehi - k];
      const int j = j1 - 1, l2 = j - q;
      const float src = cf_fc_load(c, FCo, i, j);
      const int sj1 = c.s[j1], sj = c.s[j];
      const float jb_ij = 0.0f + c.P->helix_closing[sy * 5 + sj1] + c.P->terminal_mismatch[((sy * 5 + sj1) * 5 + si1) * 5 + sj];
      const float score_other = src + jb_ij;
      out[k] = score_other + c.P->cache_single[l1 * 31 + l2] + bp_pq + jb_qp + cf_single_nuc(c, i, j, p, q);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// inside cell (i,j), InferenceEngine.ipp:3392-3688
// ---------------------------------------------------------------------------------------------
// constraint part of allow_paired for a symbol-compatible pair (a < q)
__device__ __forceinline__ bool cf_map_ok(const cf_ctx& c, int a, int q) {
  if (c.free) return true;
  const int ma = c.map[a], mq = c.map[q];
  return (ma == -1 || ma == q) && (mq == -1 || mq == a);
}

// acc (+)= term(0) (+) term(1) ... (+) term(n-1), in that order (Fast_LogPlusEquals is not associative).
// The addends are table loads from HBM/L2; sixteen of them are fetched while the previous sixteen are
// being folded, so the chain of log-sum-exps rarely waits for memory.
#define CF_FOLD 16
template <class F>
__device__ __forceinline__ float cf_fold(float acc, int n, F term) {
  int k = 0;
  if (n >= CF_FOLD) {
    float nx[CF_FOLD];
#pragma unroll
    for (int u = 0; u < CF_FOLD; ++u) nx[u] = term(u);
    for (; k + 2 * CF_FOLD <= n; k += CF_FOLD) {
      float cur[CF_FOLD];
#pragma unroll
      for (int u = 0; u < CF_FOLD; ++u) { cur[u] = nx[u]; nx[u] = term(k + CF_FOLD + u); }
#pragma unroll
      for (int u = 0; u < CF_FOLD; ++u) acc = cf_lpe(acc, cur[u]);
    }
#pragma unroll
    for (int u = 0; u < CF_FOLD; ++u) acc = cf_lpe(acc, nx[u]);
    k += CF_FOLD;
  }
  for (; k + 4 <= n; k += 4) {
    const float a0 = term(k), a1 = term(k + 1), a2 = term(k + 2), a3 = term(k + 3);
    acc = cf_lpe(cf_lpe(cf_lpe(cf_lpe(acc, a0), a1), a2), a3);
  }
  for (; k < n; ++k) acc = cf_lpe(acc, term(k));
  return acc;
}

__device__ void cf_inside_cell(const cf_ctx& c, int i, int j, float* FCi, float* FMi, float* FM1i) {
  const int L = c.L;
  const int* off = c.off;
  float FM2i = CONTRA_NEG_INF;
  if (i + 2 <= j) {
    // sequential fold over k = i+1 .. j-1 (:3392-3405)
    FM2i = cf_fold(FM2i, j - i - 1, [&](int u) { const int k = i + 1 + u; return FM1i[off[i] + k] + FMi[off[k] + j]; });
  }
  float fc = CONTRA_NEG_INF;
  const bool closing = (0 < i && j < L && cf_allow_paired(c, i, j + 1));
  if (closing) {
    float sum = CONTRA_NEG_INF;
    if (cf_all_unpaired(c, i, j)) sum = cf_lpe(sum, cf_hairpin(c, i, j));
    const float score_helix = (i + 2 <= j ? cf_base_pair(c, i + 1, j) + cf_helix_stacking(c, i, j + 1) : 0.0f);
    const float score_other = cf_junction_b(c, i, j);
    const bool pooled = c.has_pool && c.free && cf_tbase(c, c.cur)[i] >= 0;
    if (pooled) {  // the terms of the loop below, already evaluated (cf_inside_terms)
      CF_LDS const float* tl = cf_pool(c, c.cur) + cf_tbase(c, c.cur)[i];
      sum = cf_fold(sum, cf_tcnt(c, c.cur)[i], [&](int u) { return tl[u]; });
    }
    const int pmax = pooled ? i - 1 : min(i + CF_MAX_SINGLE, j);
    for (int p = i; p <= pmax; p++) {
      if (p > i && !cf_unpaired_pos(c, p)) break;
      const int q_min = max(p + 2, p - i + j - CF_MAX_SINGLE);
      const int sy = c.s[p + 1], sp = c.s[p];
      if (sy == 4) continue;  // a non-ACGU symbol pairs with nothing
      const int* pl = c.plist + sy * L;
      if (c.free) {
        // Unconstrained sequence: the same terms in the same order, software-pipelined.  A term is a chain of
        // dependent LDS lookups (partner q -> symbols and FC -> score tables -> the two lookups of the
        // log-sum-exp) and the chain, not the issue rate, is what a span waits for; so the partner of the next
        // term is fetched before this term's tables, and its symbols and FC value before this term's
        // log-sum-exp.
        int e = c.pcnt[sy * (L + 2) + j] - 1;
        if (e < 0) continue;
        int q = pl[e];
        if (q < q_min) continue;
        float inner = cf_fc_load(c, FCi, p + 1, q - 1);
        int sq_ = c.s[q], sq1 = c.s[q + 1];
        for (;;) {
          const bool more = e >= 1;
          const int qn = more ? pl[e - 1] : -1;
          const float bp = 0.0f + c.P->base_pair[sy * 5 + sq_];
          const float jb = 0.0f + c.P->helix_closing[sq_ * 5 + sy] + c.P->terminal_mismatch[((sq_ * 5 + sy) * 5 + sq1) * 5 + sp];
          const float score = (p == i && q == j)
                                  ? (score_helix + inner)
                                  : (score_other + c.P->cache_single[(p - i) * 31 + (j - q)] + inner + bp + jb + cf_single_nuc(c, i, j, p, q));
          const bool next_ok = more && qn >= q_min;
          float inner_n = 0.0f;
          int sqn = 4, sq1n = 4;
          if (next_ok) { inner_n = cf_fc_load(c, FCi, p + 1, qn - 1); sqn = c.s[qn]; sq1n = c.s[qn + 1]; }
          sum = cf_lpe(sum, score);
          if (!next_ok) break;
          q = qn; inner = inner_n; sq_ = sqn; sq1 = sq1n; --e;
        }
        continue;
      }
      for (int e = c.pcnt[sy * (L + 2) + j] - 1; e >= 0; --e) {  // partners q of p+1, q = j downwards
        const int q = pl[e];
        if (q < q_min) break;
        if (q < j && !cf_all_unpaired(c, q, j)) break;
        if (!cf_map_ok(c, p + 1, q)) continue;
        const float inner = cf_fc_load(c, FCi, p + 1, q - 1);
        // ScoreBasePair(p+1,q) and ScoreJunctionB(q,p) with the symbols of p and p+1 read once per p
        const int sq_ = c.s[q], sq1 = c.s[q + 1];
        const float bp = 0.0f + c.P->base_pair[sy * 5 + sq_];
        const float jb = 0.0f + c.P->helix_closing[sq_ * 5 + sy] + c.P->terminal_mismatch[((sq_ * 5 + sy) * 5 + sq1) * 5 + sp];
        const float score = (p == i && q == j)
                                ? (score_helix + inner)
                                : (score_other + c.P->cache_single[(p - i) * 31 + (j - q)] + inner + bp + jb + cf_single_nuc(c, i, j, p, q));
        sum = cf_lpe(sum, score);
      }
    }
    sum = cf_lpe(sum, FM2i + cf_junction_a(c, i, j) + c.P->multi_paired + c.P->multi_base);
    FCi[off[i] + j] = sum;
    fc = sum;
  }
  if (c.has_ring) c.ring[((uint32_t)(j - i) % (uint32_t)CF_RING) * (uint32_t)(L + 1) + (uint32_t)i] = fc;
  if (0 < i && i + 2 <= j && j < L) {
    float sum = CONTRA_NEG_INF;
    if (cf_allow_paired(c, i + 1, j))
      sum = cf_lpe(sum, cf_fc_load(c, FCi, i + 1, j - 1) + cf_junction_a(c, j, i) + c.P->multi_paired + cf_base_pair(c, i + 1, j));
    if (cf_unpaired_pos(c, i + 1)) sum = cf_lpe(sum, FM1i[off[i + 1] + j] + MULTI_UNPAIRED);
    FM1i[off[i] + j] = sum;
    float sm = CONTRA_NEG_INF;
    sm = cf_lpe(sm, FM2i);
    if (cf_unpaired_pos(c, j)) sm = cf_lpe(sm, FMi[off[i] + j - 1] + MULTI_UNPAIRED);
    sm = cf_lpe(sm, sum);
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
  if (a < b) {
    fmo = cf_fold(fmo, a, [&](int i) { return FM2o[off[i] + b] + FM1i[off[i] + a]; });
  }
  if (0 < a && a + 2 <= b + 1 && b + 1 < L && cf_unpaired_pos(c, b + 1)) fmo = cf_lpe(fmo, FMo[off[a] + b + 1] + MULTI_UNPAIRED);
  FMo[off[a] + b] = fmo;

  // ---- FCo[a][b] (only cells whose closing pair (a, b+1) is allowed ever receive anything)
  float fco = CONTRA_NEG_INF;
  const bool pair_ok = (0 < a && b < L && cf_allow_paired(c, a, b + 1));
  if (pair_ok) {
    const int p = a - 1, q = b + 1;
    {  // exterior loop, first sweep (:3754-3766): k = p, j = q
      const float temp = F5o[q] + c.P->external_paired + cf_base_pair(c, p + 1, q) + cf_junction_a(c, q, p);
      fco = cf_lpe(fco, temp + F5i[p]);
    }
    const float bp_pq = cf_base_pair(c, p + 1, q), jb_qp = cf_junction_b(c, q, p);  // the same two addends in every single-branch term
    const bool pooled = c.has_pool && c.free && cf_tbase(c, c.cur)[a] >= 0;
    if (pooled) {  // the partner terms of the loop below, already evaluated (cf_outside_terms); what is left is i == p's tail
      CF_LDS const float* tl = cf_pool(c, c.cur) + cf_tbase(c, c.cur)[a];
      fco = cf_fold(fco, cf_tcnt(c, c.cur)[a], [&](int u) { return tl[u]; });
    }
    for (int i = pooled ? max(1, p) : max(1, p - CF_MAX_SINGLE); i <= p; i++) {
      const int l1 = p - i;
      if (l1 > 0 && !cf_all_unpaired(c, i, p)) continue;
      const int jmax = min(L - 1, q + CF_MAX_SINGLE - l1);
      const int sy = c.s[i];
      if (sy != 4 && !pooled) {
        // sources (i,j) close the pair (i, j+1): walk the partners of i from jmax+1 down to q+2 (q+1 is the
        // (p,q) / bulge-free slot handled below for i == p, and a normal source for i < p)
        const int* pl = c.plist + sy * L;
        const int lowest = (i == p) ? q + 2 : q + 1;
        if (c.free) {
          // unconstrained: the same sources in the same order, software-pipelined like the inside loop (the next
          // partner before this term's tables, its FCo value and symbols before this term's log-sum-exp)
          int e = c.pcnt[sy * (L + 2) + jmax + 1] - 1;
          int j1 = e >= 0 ? pl[e] : 0;  // j + 1
          if (e >= 0 && j1 >= lowest) {
            const int si1 = c.s[i + 1];
            float src = cf_fc_load(c, FCo, i, j1 - 1);
            int sj1 = c.s[j1], sj = c.s[j1 - 1];
            for (;;) {
              const bool more = e >= 1;
              const int jn1 = more ? pl[e - 1] : 0;
              const int j = j1 - 1, l2 = j - q;
              const float jb_ij = 0.0f + c.P->helix_closing[sy * 5 + sj1] + c.P->terminal_mismatch[((sy * 5 + sj1) * 5 + si1) * 5 + sj];
              const float score_other = src + jb_ij;
              const float term = score_other + c.P->cache_single[l1 * 31 + l2] + bp_pq + jb_qp + cf_single_nuc(c, i, j, p, q);
              const bool next_ok = more && jn1 >= lowest;
              float src_n = 0.0f;
              int sjn1 = 4, sjn = 4;
              if (next_ok) { src_n = cf_fc_load(c, FCo, i, jn1 - 1); sjn1 = c.s[jn1]; sjn = c.s[jn1 - 1]; }
              fco = cf_lpe(fco, term);
              if (!next_ok) break;
              j1 = jn1; src = src_n; sj1 = sjn1; sj = sjn; --e;
            }
          }
        } else
        for (int e = c.pcnt[sy * (L + 2) + jmax + 1] - 1; e >= 0; --e) {
          const int j = pl[e] - 1;
          if (j + 1 < lowest) break;
          if (!cf_map_ok(c, i, j + 1)) continue;
          const int l2 = j - q;
          if (l2 > 0 && !cf_all_unpaired(c, q, j)) continue;
          const float src = cf_fc_load(c, FCo, i, j);
          const float score_other = src + cf_junction_b(c, i, j);
          fco = cf_lpe(fco, score_other + c.P->cache_single[l1 * 31 + l2] + bp_pq + jb_qp + cf_single_nuc(c, i, j, p, q));
        }
      }
      if (i == p && q <= jmax) {
        // source (p,q): block 2 (:3787-3789) comes before its own single-branch scatter (helix term)
        if (0 < p && p + 2 <= q && q < L)
          fco = cf_lpe(fco, FM1o[off[p] + q] + cf_junction_a(c, q, p) + c.P->multi_paired + cf_base_pair(c, p + 1, q));
        if (cf_allow_paired(c, p, q + 1)) {
          const float score_helix = cf_fc_load(c, FCo, p, q) + cf_base_pair(c, p + 1, q) + cf_helix_stacking(c, p, q + 1);
          fco = cf_lpe(fco, score_helix);
        }
      }
    }
    FCo[off[a] + b] = fco;
  }
  if (c.has_ring) c.ring[((uint32_t)(b - a) % (uint32_t)CF_RING) * (uint32_t)(L + 1) + (uint32_t)a] = fco;

  // ---- FM1o[a][b]: block 2 of source (a-1,b), block 4 of sources (a,j) j = L..b+1, block 1 of source (a,b)
  float fm1o = CONTRA_NEG_INF;
  if (0 < a - 1 && a - 1 + 2 <= b && b < L && cf_unpaired_pos(c, a)) fm1o = cf_lpe(fm1o, FM1o[off[a - 1] + b] + MULTI_UNPAIRED);
  if (a < b) {
    fm1o = cf_fold(fm1o, L - b, [&](int u) { const int j = L - u; return FM2o[off[a] + j] + FMi[off[b] + j]; });
  }
  const bool live = (0 < a && a + 2 <= b && b < L);
  if (live) fm1o = cf_lpe(fm1o, fmo);
  FM1o[off[a] + b] = fm1o;

  // ---- FM2o(a,b), the value the reference holds locally while scattering from (a,b)
  float fm2o = CONTRA_NEG_INF;
  if (live) fm2o = cf_lpe(fm2o, fmo);
  if (pair_ok) fm2o = cf_lpe(fm2o, fco + cf_junction_a(c, a, b) + c.P->multi_paired + c.P->multi_base);
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
      const int sy = c.s[i];
      bool slot_done = (i != p);  // the (p,q) slot carries the multi-loop term even when (p,q+1) cannot pair
      if (sy != 4) {
        const int* pl = c.plist + sy * L;
        const int e1 = c.pcnt[sy * (L + 2) + jmax + 1];
        for (int e = c.pcnt[sy * (L + 2) + q]; e < e1; ++e) {  // partners j+1 of i, j = q upwards
          const int j = pl[e] - 1;
          const int l2 = j - q;
          if (l2 > 0 && !cf_all_unpaired(c, q, j)) break;
          if (!slot_done && j > q) {  // passed the (p,q) slot without a helix term
            if (0 < p && p + 2 <= q) acc += contra_exp(FM1o[off[p] + q] + inner + cf_junction_a(c, q, p) + c.P->multi_paired + cf_base_pair(c, p + 1, q) - Z);
            slot_done = true;
          }
          if (!cf_map_ok(c, i, j + 1)) continue;
          const float outside = FCo[off[i] + j] - Z;
          float term;
          if (i == p && j == q) term = contra_exp((outside + cf_base_pair(c, i + 1, j) + cf_helix_stacking(c, i, j + 1)) + inner);
          else
            term = contra_exp((outside + cf_junction_b(c, i, j)) + c.P->cache_single[l1 * 31 + l2] + inner + cf_base_pair(c, p + 1, q) +
                              cf_junction_b(c, q, p) + cf_single_nuc(c, i, j, p, q));
          acc += term;
          if (i == p && j == q) {  // multi-loop closing pair, :4741-4745 (source (p,q), right after its single-branch block)
            if (0 < p && p + 2 <= q) acc += contra_exp(FM1o[off[p] + q] + inner + cf_junction_a(c, q, p) + c.P->multi_paired + cf_base_pair(c, p + 1, q) - Z);
            slot_done = true;
          }
        }
      }
      if (!slot_done) {
        if (0 < p && p + 2 <= q) acc += contra_exp(FM1o[off[p] + q] + inner + cf_junction_a(c, q, p) + c.P->multi_paired + cf_base_pair(c, p + 1, q) - Z);
      }
    }
  }
  // exterior, :4750-4760
  acc += contra_exp((F5o[q] - Z) + F5i[p] + inner + c.P->external_paired + cf_base_pair(c, p + 1, q) + cf_junction_a(c, q, p));
  const float m = acc < 0.0f ? 0.0f : acc;  // Clip, Utilities.ipp:136
  return (1.0f < m) ? 1.0f : m;
}

// ---------------------------------------------------------------------------------------------
// integer side tables of one sequence, in LDS: s, map, cum, off (L+2 each), plist (4*L), pcnt (4*(L+2))
#define CF_INTS(L) (12 * ((L) + 2))

__device__ void cf_bind(cf_ctx& c, int L, int* ints, float* ring, const cf_params* P) {
  c.has_ring = ring != nullptr;
  c.ring = (CF_LDS float*)ring;
  c.L = L;
  int* s = ints;
  int* map = s + (L + 2);
  int* cum = map + (L + 2);
  int* off = cum + (L + 2);
  int* plist = off + (L + 2);
  int* pcnt = plist + 4 * (L + 2);
  c.s = s; c.map = map; c.cum = cum; c.off = off; c.plist = plist; c.pcnt = pcnt; c.P = P;
}

// pool_floats: capacity of the term pool in the dynamic LDS behind the ring (0 = no pool); cell_waves: the cells of a span
// are dealt to this many wavefronts (the chains are latency-bound and every wavefront they are spread over adds its
// whole instruction stream to the SIMD's issue load), while the term evaluation uses all of them
// pool_bufs: 1 = one pool, filled by every wavefront before the cells of a span run; 2 = two halves, the next span's terms
// evaluated by the wavefronts without cells while this span's cells run
__global__ __launch_bounds__(CF_FOLD_THREADS) void k_contrafold(cf_batch B, int use_ring, int pool_floats, int cell_waves, int pool_bufs) {
  CF_TABLES_INIT();
  __shared__ cf_params sP;
  __shared__ float s_terms[CF_FOLD_THREADS];
  {
    const float* src = (const float*)B.params;
    float* dst = (float*)&sP;
    for (uint32_t k = threadIdx.x; k < sizeof(cf_params) / 4; k += blockDim.x) dst[k] = src[k];
  }
  const uint32_t x = blockIdx.x;
  const cf_seq sq = B.seqs[x];
  const int L = (int)sq.len;
  const int tid = threadIdx.x, nt = blockDim.x;
  const int cw = min(cell_waves, nt >> 6);
  const int cnt_ = cw * 64;                                            // threads that own cells
  const int cid = (tid >> 6) < cw ? (tid & 63) * cw + (tid >> 6) : nt;  // this thread's place in a span (see CF_FOLD_THREADS); nt = none
  extern __shared__ int s_ints[];
  float* ring = use_ring ? (float*)(s_ints + CF_INTS(L)) : nullptr;
  cf_ctx c;
  cf_bind(c, L, s_ints, ring, &sP);
  c.free = !sq.has_constraint;
  {
    int* after = s_ints + CF_INTS(L) + (use_ring ? CF_RING * (L + 1) : 0);
    c.has_pool = pool_floats > 0 && c.free;
    c.tbase = (CF_LDS int*)after;
    c.tcnt = (CF_LDS int*)(after + 2 * (L + 1));
    c.ptop = (CF_LDS int*)(after + 4 * (L + 1));
    c.pool = (CF_LDS float*)(after + 4 * (L + 1) + 2);
    c.pool_cap = pool_floats / pool_bufs;
    c.cur = 0;
    if (tid == 0 && c.has_pool) c.ptop[0] = c.ptop[1] = 0;
  }
  const bool overlap = c.has_pool && pool_bufs == 2 && cnt_ < nt;  // wavefronts left over for the terms
  const int tw0 = overlap ? cnt_ : 0, twn = nt - tw0;               // the threads that evaluate terms inside the span loops
  int* s = (int*)c.s; int* map = (int*)c.map; int* cum = (int*)c.cum; int* off = (int*)c.off;
  int* plist = (int*)c.plist; int* pcnt = (int*)c.pcnt;
  float* F = B.fws + sq.fws_off;
  const size_t SZ = (size_t)(L + 1) * (L + 2) / 2;
  float *FCi = F, *FMi = F + SZ, *FM1i = F + 2 * SZ, *FCo = F + 3 * SZ, *FMo = F + 4 * SZ, *FM1o = F + 5 * SZ, *FM2o = F + 6 * SZ;
  float *F5i = F + 7 * SZ, *F5o = F5i + (L + 1);

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
  if (ring)
    for (int k = tid; k < CF_RING * (L + 1); k += nt) ring[k] = CONTRA_NEG_INF;
  __syncthreads();
  if (tid == 0) {
    int run = 0;
    cum[0] = 0;
    for (int i = 1; i <= L; ++i) { run += (map[i] == -1 || map[i] == 0) ? 0 : 1; cum[i] = run; }
    cum[L + 1] = run;
  }
  if (tid >= 64 && tid < 68) {  // partner list of symbol y: positions whose symbol pairs with y
    const int y = tid - 64;
    int n = 0;
    pcnt[y * (L + 2)] = 0;
    for (int q = 1; q <= L; ++q) {
      if (cf_comp(y, s[q])) plist[y * L + n++] = q;
      pcnt[y * (L + 2) + q] = n;
    }
    pcnt[y * (L + 2) + L + 1] = n;
  }
  __syncthreads();
  // keep the integer tables for the posterior kernel
  for (int k = tid; k < CF_INTS(L); k += nt) B.iws[sq.iws_off + k] = s_ints[k];

#define CF_STAMP(k) if (B.stamps && blockIdx.x == 0 && tid == 0) B.stamps[k] = wall_clock64()
  if (B.stamps && blockIdx.x == 0 && tid == 0) B.stamps[5] = B.stamps[6] = 0;
  CF_STAMP(0);
  // ---- inside: span ascending
  if (overlap) {  // the first span's terms (none: spans below 2 have no single-branch term, but the lists must exist)
    cf_inside_terms(c, 0, FCi, 0, 0, nt);
    __syncthreads();
  }
  for (int d = 0; d <= L; ++d) {
    if (overlap) {
      // cells of span d from buffer d & 1; meanwhile the other wavefronts evaluate span d + 1 (its terms read FC of spans
      // d - 1 and below, and the ring slot span d overwrites held span d - 33) into the other buffer, whose bump counter
      // was cleared during span d - 1
      c.cur = d & 1;
      if (tid == 0) c.ptop[d & 1] = 0;  // for span d + 2; nothing bumps or reads this counter now
      if (tid >= tw0) {
        if (d + 1 <= L) cf_inside_terms(c, d + 1, FCi, (d + 1) & 1, tw0, twn);
      } else {
        for (int i = cid; i + d <= L; i += cnt_) cf_inside_cell(c, i, i + d, FCi, FMi, FM1i);
      }
      __syncthreads();
      continue;
    }
    if (c.has_pool) {
      const unsigned long long t0 = (B.stamps && blockIdx.x == 0 && tid == 0) ? wall_clock64() : 0;
      cf_inside_terms(c, d, FCi, 0, 0, nt);
      __syncthreads();
      if (B.stamps && blockIdx.x == 0 && tid == 0) B.stamps[5] += wall_clock64() - t0;
    }
    if (tid == 0 && c.has_pool) c.ptop[0] = 0;  // every bump of this span is behind the barrier, the next span's in front of the one below
    for (int i = cid; i + d <= L; i += cnt_) cf_inside_cell(c, i, i + d, FCi, FMi, FM1i);
    __syncthreads();
  }
  CF_STAMP(1);
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
        if (k0 == 0 && cf_unpaired_pos(c, j)) sum = cf_lpe(sum, F5i[j - 1] + EXT_UNPAIRED);
        const int n = min(nt, j - k0);
        for (int u = 0; u < n; ++u) {
          const float v = s_terms[u];
          if (v == v) sum = cf_lpe(sum, v);
        }
        F5i[j] = sum;
      }
      __syncthreads();
    }
  }
  const float Z = F5i[L];
  CF_STAMP(2);

  // ---- outside.  F5o first (:3746-3767): lane k owns F5o[k]; at step tau all lanes take the
  // addend of j = L - tau + 1, whose F5o[j] was completed in the previous steps.
  if (tid == 0) F5o[L] = 0.0f;
  if (ring)
    for (int k = tid; k < CF_RING * (L + 1); k += nt) ring[k] = CONTRA_NEG_INF;
  __syncthreads();
  for (int j = L; j >= 1; --j) {
    const float f5oj = F5o[j];
    for (int k = cid; k < j; k += cnt_) {
      float v = F5o[k];
      if (k == j - 1 && cf_unpaired_pos(c, j)) v = cf_lpe(v, f5oj + EXT_UNPAIRED);
      if (cf_allow_paired(c, k + 1, j)) {
        const float temp = f5oj + sP.external_paired + cf_base_pair(c, k + 1, j) + cf_junction_a(c, j, k);
        v = cf_lpe(v, temp + FCi[off[k + 1] + j - 1]);
      }
      F5o[k] = v;
    }
    __syncthreads();
  }
  CF_STAMP(3);
  // main sweep: span descending
  if (tid == 0 && c.has_pool) c.ptop[0] = c.ptop[1] = 0;
  __syncthreads();
  if (overlap) {
    cf_outside_terms(c, L, FCo, L & 1, 0, nt);
    __syncthreads();
  }
  for (int d = L; d >= 0; --d) {
    if (overlap) {  // as in the inside pass: span d - 1's sources are FCo of spans d + 1 and above
      c.cur = d & 1;
      if (tid == 0) c.ptop[d & 1] = 0;
      if (tid >= tw0) {
        if (d - 1 >= 0) cf_outside_terms(c, d - 1, FCo, (d - 1) & 1, tw0, twn);
      } else {
        for (int a = cid; a + d <= L; a += cnt_) cf_outside_cell(c, a, a + d, FCi, FMi, FM1i, F5i, F5o, FCo, FMo, FM1o, FM2o);
      }
      __syncthreads();
      continue;
    }
    if (c.has_pool) {
      const unsigned long long t0 = (B.stamps && blockIdx.x == 0 && tid == 0) ? wall_clock64() : 0;
      cf_outside_terms(c, d, FCo, 0, 0, nt);
      __syncthreads();
      if (B.stamps && blockIdx.x == 0 && tid == 0) B.stamps[6] += wall_clock64() - t0;
    }
    if (tid == 0 && c.has_pool) c.ptop[0] = 0;
    for (int a = cid; a + d <= L; a += cnt_) cf_outside_cell(c, a, a + d, FCi, FMi, FM1i, F5i, F5o, FCo, FMo, FM1o, FM2o);
    __syncthreads();
  }
  CF_STAMP(4);
  if (tid == 0 && B.logz) B.logz[x] = Z;
}

// posterior: one workgroup per (sequence, row a); fully parallel over pairs
__global__ __launch_bounds__(CF_THREADS) void k_contrafold_posterior(cf_batch B) {
  __shared__ cf_params sP;
  {
    const float* src = (const float*)B.params;
    float* dst = (float*)&sP;
    for (uint32_t k = threadIdx.x; k < sizeof(cf_params) / 4; k += blockDim.x) dst[k] = src[k];
  }
  const cf_seq sq = B.seqs[blockIdx.y];
  const int L = (int)sq.len;
  const int a = (int)blockIdx.x;  // row 0..L (row 0 and the diagonal stay 0)
  if (a > L) return;
  extern __shared__ int s_ints[];
  for (int k = threadIdx.x; k < CF_INTS(L); k += blockDim.x) s_ints[k] = B.iws[sq.iws_off + k];
  __syncthreads();
  cf_ctx c;
  cf_bind(c, L, s_ints, nullptr, &sP);
  c.free = !sq.has_constraint;
  float* F = B.fws + sq.fws_off;
  const size_t SZ = (size_t)(L + 1) * (L + 2) / 2;
  const float *FCi = F, *FCo = F + 3 * SZ, *FM1o = F + 5 * SZ, *F5i = F + 7 * SZ, *F5o = F5i + (L + 1);
  float* post = B.post + sq.post_off;
  const float Z = F5i[L];
  for (int q = a + (int)threadIdx.x; q <= L; q += blockDim.x)
    post[c.off[a] + q] = (a >= 1 && q > a) ? cf_posterior_cell(c, a, q, FCi, F5i, F5o, FCo, FM1o, Z) : 0.0f;
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
  const size_t ints = (size_t)CF_INTS(max_len) * sizeof(int);
  const size_t ring = (size_t)CF_RING * (max_len + 1) * sizeof(float);
  const size_t budget = 100 * 1024;  // dynamic LDS (static: score tables ~11 KB)
  if (ints > budget) return DAFS_HIP_ETOOLONG;
  // What the CU's LDS is split into, behind the kernel's static tables: the integer side tables (always), the term pool and
  // the ring of recent FC spans.  The pool comes first: it is what takes the term evaluation out of the chains; the
  // ring only shortens the FC reads of that evaluation, which the whole workgroup issues in parallel anyway.  So the
  // ring is kept when a pool of ~80 terms per row (the demand of a random sequence: 3/8 of the cells close a pair,
  // ~150 terms each) still fits beside it, and dropped otherwise (from ~330 nt on).
  size_t stat = 28 * 1024;  // static LDS of k_contrafold (score tables, log-sum-exp tables)
  const size_t total = 160 * 1024 - 512;
  {
    hipFuncAttributes at;
    if (hipFuncGetAttributes(&at, (const void*)k_contrafold) == hipSuccess) stat = at.sharedSizeBytes;
    else (void)hipGetLastError();
  }
  const size_t pool_ints = (4 * ((size_t)max_len + 1) + 2) * sizeof(int);
  const size_t pool_want = (size_t)80 * (max_len + 1) * sizeof(float);
  const bool pool_on = !getenv("DAFS_HIP_CF_NOPOOL") && stat + ints + pool_ints + 2 * 496 * sizeof(float) <= total;
  int use_ring = ints + ring <= budget && !getenv("DAFS_HIP_CF_NORING");  // the env switches are tuning aids
  if (use_ring && pool_on && stat + ints + ring + pool_ints + pool_want > total) use_ring = 0;
  size_t lds = ints + (use_ring ? ring : 0);
  int pool_floats = 0;
  if (pool_on) {
    const size_t fixed = lds + pool_ints;
    pool_floats = (int)((total - stat - fixed) / sizeof(float));
    lds = fixed + (size_t)pool_floats * sizeof(float);
  }
  // two halves (terms of the next span evaluated beside the cells of this one) when each still holds the typical demand
  int pool_bufs = (pool_floats / 2 >= (int)(pool_want / sizeof(float)) && !getenv("DAFS_HIP_CF_NOOVERLAP")) ? 2 : 1;
  int cell_waves = 8;
  if (const char* e = getenv("DAFS_HIP_CF_CELL_WAVES")) {  // tuning aid
    const int v = atoi(e);
    if (v >= 1 && v <= CF_FOLD_THREADS / 64) cell_waves = v;
  }
  const size_t optin = total - stat;  // the most dynamic LDS a launch may ask for
  static bool attr_a[16] = {false}, attr_b[16] = {false};
  if (!lds_optin_once((const void*)k_contrafold, (int)optin, attr_a)) return DAFS_HIP_ELAUNCH;
  if (!lds_optin_once((const void*)k_contrafold_posterior, (int)budget, attr_b)) return DAFS_HIP_ELAUNCH;
  int fold_threads = CF_FOLD_THREADS;
  if (const char* e = getenv("DAFS_HIP_CF_THREADS")) {  // tuning aid: 64..1024 in whole wavefronts
    const int v = atoi(e);
    if (v >= 64 && v <= CF_FOLD_THREADS && v % 64 == 0) fold_threads = v;
  }
  STAGE_LAUNCH(ST_CONTRAFOLD, st) hipLaunchKernelGGL(k_contrafold, dim3(nseq), dim3(fold_threads), lds, st, B, use_ring, pool_floats, cell_waves, pool_bufs);
  if (hip_check(hipGetLastError())) return DAFS_HIP_ELAUNCH;
  STAGE_LAUNCH(ST_CF_POSTERIOR, st) hipLaunchKernelGGL(k_contrafold_posterior, dim3(max_len + 1, nseq), dim3(CF_THREADS), ints, st, B);
  return hip_check(hipGetLastError()) ? DAFS_HIP_ELAUNCH : DAFS_HIP_OK;
}

int bp_compact_launch(const cf_batch& B, uint32_t nseq, float th, const uint64_t* rp_off, uint32_t* out_rowptr, uint32_t* out_col,
                      float* out_val, uint64_t* out_off, uint32_t* out_nnz, unsigned long long* pool_top, uint64_t pool_cap, int* status,
                      hipStream_t st) {
  if (!nseq) return DAFS_HIP_OK;
  STAGE_LAUNCH(ST_BP_COMPACT, st) hipLaunchKernelGGL(k_bp_compact, dim3(nseq), dim3(256), 0, st, B, th, rp_off, out_rowptr, out_col, out_val, out_off, out_nnz, pool_top,
                     pool_cap, status);
  return hip_check(hipGetLastError()) ? DAFS_HIP_ELAUNCH : DAFS_HIP_OK;
}

}  // namespace dafs
