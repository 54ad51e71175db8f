/* oracle/pipeline.c -- TEST INFRASTRUCTURE (see oracle.h).
 *
 * CPU restatement of the parts of src/dafs.cpp that call, or sit between, the hot-path
 * plugins: transpose_mp, calculate_similarity_score, the two probabilistic-consistency
 * transforms, build_tree/print_tree, posterior averaging, solve_by_dd, the projections,
 * the progressive recursion and the output format.
 *
 * PINNING.  src/dafs.cpp itself cannot be compiled in this image (it includes ViennaRNA,
 * cxxopts and spdlog headers, all absent; writing stand-ins is not allowed), so this file is
 * pinned by known answers only:
 *   - README.md:59  guide-tree line for examples/RF00005:0.fa (pins ProbCons -> MP -> sim -> tree -> print)
 *   - SURVEY.md Appendix C (captured from the reference during the survey): guide trees for
 *     RF00005:0 / RF00017:4, and for `-s CONTRAfold --no-alifold` on RF00005:0 the column count
 *     (86) and the first / last alignment rows (pins PCT, averaging, DD loop, projections).
 * The >SS_cons line of the reference always contains an RNAalifold term (dafs.cpp:82,1862)
 * that needs ViennaRNA: parity unpinned for that line; here it is the plain Nussinov decode
 * of the averaged base-pairing matrix with no alifold term.
 */
#include "oracle.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#define CUTOFF 0.01 /* dafs.cpp:65 (double literal, compared against floats as double) */

typedef struct {
  uint32_t n, L;
  uint32_t* idx; /* sequence index per row */
  uint8_t* mask; /* n*L, 1 = residue, 0 = gap (vector<bool> in the reference) */
} aln_t;

struct orc_pipeline {
  orc_params prm;
  uint32_t N;
  char** names;
  char** seqs;
  uint32_t* len;
  orc_csr* bp;  /* N */
  orc_csr* mp;  /* N*N */
  float* sim;   /* N*N */
  float* tscore; uint32_t* tleft; uint32_t* tright; /* 2N-1 */
  aln_t final_aln;
  uint32_t* final_ss; char* final_str;
  char* out; size_t out_len, out_cap;
  uint32_t* dd_iters; uint32_t* dd_viol; uint32_t dd_n;
  double secs[4];
};

static double now_s(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

void orc_params_default(orc_params* p) { /* dafs.cpp:1612-1640 */
  p->align_model = 0; p->fold_model = 0;
  p->w = 4.0f; p->eta0 = 0.5f; p->t_max = 600;
  p->w_pct_a = 0.25f; p->w_pct_s = 0.25f; p->th_a = 0.01f; p->th_s = 0.2f; p->th_s1 = 0.2f;
  p->force_iters = 0;
  p->w_pct_f = 0.0f; p->bp_update = 0; p->bp_update1 = 0;
}

static void csr_alloc(orc_csr* m, uint32_t nrow, uint32_t cap) {
  m->nrow = nrow;
  m->rowptr = (uint32_t*)calloc((size_t)nrow + 1, sizeof(uint32_t));
  m->col = (uint32_t*)malloc(((size_t)cap + 1) * sizeof(uint32_t));
  m->val = (float*)malloc(((size_t)cap + 1) * sizeof(float));
}

/* transpose_mp, dafs.cpp:155-167: rows of the transpose sorted by (i, p) -- i is unique per row
 * so a counting transpose in row order gives the same lists. */
void orc_transpose(const orc_csr* in, uint32_t ncol, orc_csr* out) {
  uint32_t nnz = in->rowptr[in->nrow];
  csr_alloc(out, ncol, nnz);
  for (uint32_t e = 0; e < nnz; ++e) out->rowptr[in->col[e] + 1]++;
  for (uint32_t j = 0; j < ncol; ++j) out->rowptr[j + 1] += out->rowptr[j];
  uint32_t* cur = (uint32_t*)malloc(((size_t)ncol + 1) * sizeof(uint32_t));
  memcpy(cur, out->rowptr, ((size_t)ncol + 1) * sizeof(uint32_t));
  for (uint32_t i = 0; i < in->nrow; ++i)
    for (uint32_t e = in->rowptr[i]; e < in->rowptr[i + 1]; ++e) {
      uint32_t pos = cur[in->col[e]]++;
      out->col[pos] = i;
      out->val[pos] = in->val[e];
    }
  free(cur);
}

/* calculate_similarity_score, dafs.cpp:713-764 */
float orc_similarity_score(const uint32_t* rowptr, const uint32_t* col, const float* val, uint32_t L1, uint32_t L2) {
  const size_t W = (size_t)L2 + 1;
  float* dp = (float*)calloc((size_t)(L1 + 1) * W, sizeof(float));
  int* tr = (int*)calloc((size_t)(L1 + 1) * W, sizeof(int));
  for (uint32_t i = 1; i != L1 + 1; ++i) {
    uint32_t j = 1;
    for (uint32_t e = rowptr[i - 1]; e < rowptr[i]; ++e) {
      for (; j - 1 < col[e]; ++j) {
        dp[i * W + j] = dp[i * W + j - 1];
        tr[i * W + j] = tr[i * W + j - 1] + 1;
        if (dp[i * W + j] < dp[(i - 1) * W + j]) {
          dp[i * W + j] = dp[(i - 1) * W + j];
          tr[i * W + j] = tr[(i - 1) * W + j] + 1;
        }
      }
      dp[i * W + j] = dp[(i - 1) * W + j - 1] + val[e];
      tr[i * W + j] = tr[(i - 1) * W + j - 1] + 1;
      if (dp[i * W + j] < dp[i * W + j - 1]) {
        dp[i * W + j] = dp[i * W + j - 1];
        tr[i * W + j] = tr[i * W + j - 1] + 1;
      }
      if (dp[i * W + j] < dp[(i - 1) * W + j]) {
        dp[i * W + j] = dp[(i - 1) * W + j];
        tr[i * W + j] = tr[(i - 1) * W + j] + 1;
      }
      ++j;
    }
    for (; j < L2 + 1; ++j) {
      dp[i * W + j] = dp[i * W + j - 1];
      tr[i * W + j] = tr[i * W + j - 1] + 1;
      if (dp[i * W + j] < dp[(i - 1) * W + j]) {
        dp[i * W + j] = dp[(i - 1) * W + j];
        tr[i * W + j] = tr[(i - 1) * W + j] + 1;
      }
    }
  }
  float r = dp[(size_t)L1 * W + L2] / tr[(size_t)L1 * W + L2];
  free(dp); free(tr);
  return r;
}

#define MP(pl, x, y) (&(pl)->mp[(size_t)(x) * (pl)->N + (y)])
#define SIM(pl, x, y) ((pl)->sim[(size_t)(x) * (pl)->N + (y)])

static void csr_identity(orc_csr* m, uint32_t L) { /* align.cpp:42-44, dafs.cpp:317-322 */
  csr_alloc(m, L, L);
  for (uint32_t x = 0; x < L; ++x) { m->rowptr[x] = x; m->col[x] = x; m->val[x] = 1.0f; }
  m->rowptr[L] = L;
}

/* dense -> rows with v > CUTOFF (dafs.cpp:303-312 / :365-372) */
static void dense_to_csr(orc_csr* m, const float* d, uint32_t R, uint32_t Cn, float sum_w, int upper_only) {
  uint32_t nnz = 0;
  for (uint32_t i = 0; i < R; ++i)
    for (uint32_t j = upper_only ? i + 1 : 0; j < Cn; ++j) {
      float v = d[(size_t)i * Cn + j] / sum_w;
      if (v > CUTOFF) nnz++;
    }
  csr_alloc(m, R, nnz);
  uint32_t n = 0;
  for (uint32_t i = 0; i < R; ++i) {
    m->rowptr[i] = n;
    for (uint32_t j = upper_only ? i + 1 : 0; j < Cn; ++j) {
      float v = d[(size_t)i * Cn + j] / sum_w;
      if (v > CUTOFF) { m->col[n] = j; m->val[n] = v; ++n; }
    }
  }
  m->rowptr[R] = n;
}

/* DAFS::relax_matching_probability, dafs.cpp:258-324 */
static void relax_matching_probability(orc_pipeline* pl) {
  const uint32_t N = pl->N;
  const float w_pct_a = pl->prm.w_pct_a;
  orc_csr* mp = (orc_csr*)calloc((size_t)N * N, sizeof(orc_csr));
  for (uint32_t x = 0; x + 1 < N; ++x) {
    const uint32_t L1 = pl->len[x];
    for (uint32_t y = x + 1; y != N; ++y) {
      const uint32_t L2 = pl->len[y];
      float* posterior = (float*)calloc((size_t)L1 * L2, sizeof(float));
      float sum_w = 0.0;
      for (uint32_t z = 0; z != N; ++z) {
        const uint32_t L3 = pl->len[z];
        float w = SIM(pl, z, x) * SIM(pl, z, y);
        if (w_pct_a < 0.0) w *= 1.0 / N;
        else if (z == x || z == y) w *= (1.0 - w_pct_a) / 2;
        else w *= w_pct_a / (N - 2);
        sum_w += w;
        const orc_csr* zx = MP(pl, z, x);
        const orc_csr* zy = MP(pl, z, y);
        for (uint32_t k = 0; k != L3; ++k)
          for (uint32_t a = zx->rowptr[k]; a < zx->rowptr[k + 1]; ++a) {
            const uint32_t i = zx->col[a];
            const float p_ik = zx->val[a];
            for (uint32_t b = zy->rowptr[k]; b < zy->rowptr[k + 1]; ++b)
              posterior[(size_t)i * L2 + zy->col[b]] += p_ik * zy->val[b] * w;
          }
      }
      dense_to_csr(&mp[(size_t)x * N + y], posterior, L1, L2, sum_w, 0);
      orc_transpose(&mp[(size_t)x * N + y], L2, &mp[(size_t)y * N + x]);
      free(posterior);
    }
  }
  for (uint32_t x = 0; x != N; ++x) csr_identity(&mp[(size_t)x * N + x], pl->len[x]);
  for (size_t e = 0; e < (size_t)N * N; ++e) orc_csr_free(&pl->mp[e]);
  free(pl->mp);
  pl->mp = mp;
}

/* DAFS::relax_fourway_consistency, dafs.cpp:377-444: mp_[x][y] is replaced by a mix of itself and the stacking
 * evidence of the two base-pairing matrices (the loops, their order and the types of every product are the
 * reference's: p_ik * (1.0 - w) is a double product added to a float cell, the other addends are float products). */
static void relax_fourway_consistency(orc_pipeline* pl) {
  const uint32_t N = pl->N;
  const float w_pct_f = pl->prm.w_pct_f;
  orc_csr* mp = (orc_csr*)calloc((size_t)N * N, sizeof(orc_csr));
  for (uint32_t x = 0; x + 1 < N; ++x) {
    const uint32_t L1 = pl->len[x];
    const orc_csr* bx = &pl->bp[x];
    for (uint32_t y = x + 1; y != N; ++y) {
      const uint32_t L2 = pl->len[y];
      const orc_csr* by = &pl->bp[y];
      const orc_csr* m = MP(pl, x, y);
      float* posterior = (float*)calloc((size_t)L1 * L2, sizeof(float));
      for (uint32_t i = 0; i != L1; ++i) {
        for (uint32_t a = m->rowptr[i]; a < m->rowptr[i + 1]; ++a) {
          const uint32_t k = m->col[a];
          const float p_ik = m->val[a];
          posterior[(size_t)i * L2 + k] += p_ik * (1.0 - w_pct_f);
          for (uint32_t b = bx->rowptr[i]; b < bx->rowptr[i + 1]; ++b) {
            const uint32_t j = bx->col[b];
            const float p_ij = bx->val[b];
            uint32_t l1 = m->rowptr[j], e1 = m->rowptr[j + 1];
            uint32_t l2 = by->rowptr[k], e2 = by->rowptr[k + 1];
            while (l1 != e1 && l2 != e2) {
              if (m->col[l1] < by->col[l2]) ++l1;
              else if (m->col[l1] > by->col[l2]) ++l2;
              else {
                const uint32_t l = m->col[l1];
                const float p_jl = m->val[l1];
                const float p_kl = by->val[l2];
                posterior[(size_t)i * L2 + k] += p_ij * p_kl * p_jl * w_pct_f;
                posterior[(size_t)j * L2 + l] += p_ij * p_kl * p_ik * w_pct_f;
                ++l1;
                ++l2;
              }
            }
          }
        }
      }
      dense_to_csr(&mp[(size_t)x * N + y], posterior, L1, L2, 1.0f, 0); /* v > CUTOFF, :428-436 (x / 1.0f is x) */
      orc_transpose(&mp[(size_t)x * N + y], L2, &mp[(size_t)y * N + x]);
      free(posterior);
    }
  }
  for (uint32_t x = 0; x != N; ++x) csr_identity(&mp[(size_t)x * N + x], pl->len[x]);
  for (size_t e = 0; e < (size_t)N * N; ++e) orc_csr_free(&pl->mp[e]);
  free(pl->mp);
  pl->mp = mp;
}

/* DAFS::relax_basepairing_probability, dafs.cpp:326-375 */
static void relax_basepairing_probability(orc_pipeline* pl) {
  const uint32_t N = pl->N;
  const float w_pct_s = pl->prm.w_pct_s;
  orc_csr* bp = (orc_csr*)calloc(N, sizeof(orc_csr));
  for (uint32_t x = 0; x != N; ++x) {
    const uint32_t L1 = pl->len[x];
    float* p = (float*)calloc((size_t)L1 * L1, sizeof(float));
    float sum_w = 0.0;
    for (uint32_t y = 0; y != N; ++y) {
      const uint32_t L2 = pl->len[y];
      float w = SIM(pl, y, x);
      if (w_pct_s < 0.0) w *= 1.0 / N;
      else if (y == x) w *= 1.0 - w_pct_s;
      else w *= w_pct_s / (N - 1);
      sum_w += w;
      const orc_csr* b = &pl->bp[y];
      const orc_csr* m = MP(pl, y, x);
      for (uint32_t k = 0; k != L2; ++k)
        for (uint32_t e = b->rowptr[k]; e < b->rowptr[k + 1]; ++e) {
          const uint32_t l = b->col[e];
          const float p_kl = b->val[e];
          for (uint32_t a = m->rowptr[k]; a < m->rowptr[k + 1]; ++a) {
            const uint32_t i = m->col[a];
            const float p_ik = m->val[a];
            for (uint32_t c = m->rowptr[l]; c < m->rowptr[l + 1]; ++c) {
              const uint32_t j = m->col[c];
              if (i < j) p[(size_t)i * L1 + j] += p_kl * p_ik * m->val[c] * w;
            }
          }
        }
    }
    /* :365-372: rows i in [0,L1-1), j>i */
    dense_to_csr(&bp[x], p, L1, L1, sum_w, 1);
    free(p);
  }
  for (uint32_t x = 0; x < N; ++x) orc_csr_free(&pl->bp[x]);
  free(pl->bp);
  pl->bp = bp;
}

/* DAFS::build_tree, dafs.cpp:446-492.  std::priority_queue<pair<float,pair<uint,uint>>> is a
 * max-heap under lexicographic operator<; emulated with a binary heap using the same order. */
typedef struct { float s; uint32_t a, b; } hnode;
static int hless(const hnode* x, const hnode* y) {
  if (x->s < y->s) return 1;
  if (y->s < x->s) return 0;
  if (x->a < y->a) return 1;
  if (y->a < x->a) return 0;
  return x->b < y->b;
}
static void hpush(hnode* h, size_t* n, hnode v) {
  size_t i = (*n)++;
  h[i] = v;
  while (i > 0) {
    size_t p = (i - 1) / 2;
    if (hless(&h[p], &h[i])) { hnode t = h[p]; h[p] = h[i]; h[i] = t; i = p; }
    else break;
  }
}
static hnode hpop(hnode* h, size_t* n) {
  hnode top = h[0];
  h[0] = h[--(*n)];
  size_t i = 0;
  for (;;) {
    size_t l = 2 * i + 1, r = l + 1, m = i;
    if (l < *n && hless(&h[m], &h[l])) m = l;
    if (r < *n && hless(&h[m], &h[r])) m = r;
    if (m == i) break;
    hnode t = h[m]; h[m] = h[i]; h[i] = t; i = m;
  }
  return top;
}

static void build_tree(orc_pipeline* pl) {
  uint32_t n = pl->N;
  const uint32_t n0 = n, T = 2 * n - 1;
  for (uint32_t i = 0; i < T; ++i) { pl->tscore[i] = 0.0f; pl->tleft[i] = ORC_NONE; pl->tright[i] = ORC_NONE; }
  float* d = (float*)calloc((size_t)n0 * n0, sizeof(float));
  uint32_t* idx = (uint32_t*)malloc(T * sizeof(uint32_t));
  for (uint32_t i = 0; i < T; ++i) idx[i] = i < n0 ? i : ORC_NONE;
  hnode* h = (hnode*)malloc(((size_t)n0 * n0 + 4 * (size_t)n0 * n0 + 16) * sizeof(hnode));
  size_t hn = 0;
  for (uint32_t i = 0; i + 1 < n; ++i)
    for (uint32_t j = i + 1; j != n; ++j) {
      d[(size_t)i * n0 + j] = d[(size_t)j * n0 + i] = SIM(pl, i, j);
      hnode v = {SIM(pl, i, j), i, j};
      hpush(h, &hn, v);
    }
  while (hn) {
    hnode t = hpop(h, &hn);
    if (idx[t.a] != ORC_NONE && idx[t.b] != ORC_NONE) {
      const uint32_t l = idx[t.a], r = idx[t.b];
      idx[t.a] = idx[t.b] = ORC_NONE;
      for (uint32_t i = 0; i != n; ++i) {
        if (idx[i] != ORC_NONE) {
          uint32_t ii = idx[i];
          float v = (d[(size_t)ii * n0 + l] + d[(size_t)ii * n0 + r]) * t.s / 2;
          d[(size_t)ii * n0 + l] = d[(size_t)l * n0 + ii] = v;
          hnode nv = {v, i, n};
          hpush(h, &hn, nv);
        }
      }
      pl->tscore[n] = t.s; pl->tleft[n] = t.a; pl->tright[n] = t.b;
      idx[n++] = l;
    }
  }
  free(d); free(idx); free(h);
}

/* ---------------- output buffer ---------------- */
static void out_put(orc_pipeline* pl, const char* s, size_t n) {
  if (pl->out_len + n + 1 > pl->out_cap) {
    pl->out_cap = (pl->out_len + n + 1) * 2 + 256;
    pl->out = (char*)realloc(pl->out, pl->out_cap);
  }
  memcpy(pl->out + pl->out_len, s, n);
  pl->out_len += n;
  pl->out[pl->out_len] = 0;
}
static void out_str(orc_pipeline* pl, const char* s) { out_put(pl, s, strlen(s)); }

/* print_tree, dafs.cpp:495-511; operator<<(float) == "%g" */
static void print_tree(orc_pipeline* pl, uint32_t i) {
  if (pl->tleft[i] == ORC_NONE) out_str(pl, pl->names[i]);
  else {
    char buf[64];
    snprintf(buf, sizeof buf, "[ %g ", (double)pl->tscore[i]);
    out_str(pl, buf);
    print_tree(pl, pl->tleft[i]);
    out_str(pl, " ");
    print_tree(pl, pl->tright[i]);
    out_str(pl, " ]");
  }
}

/* ---------------- phase 2 ---------------- */
static void aln_free(aln_t* a) { free(a->idx); free(a->mask); a->idx = NULL; a->mask = NULL; }

/* average_matching_probability, dafs.cpp:513-559 */
static float* average_matching_probability(const orc_pipeline* pl, const aln_t* a1, const aln_t* a2) {
  const uint32_t L1 = a1->L, L2 = a2->L, N1 = a1->n, N2 = a2->n;
  float* p = (float*)calloc((size_t)L1 * L2, sizeof(float));
  for (uint32_t r1 = 0; r1 < N1; ++r1)
    for (uint32_t r2 = 0; r2 < N2; ++r2) {
      const orc_csr* m = MP(pl, a1->idx[r1], a2->idx[r2]);
      const uint8_t* m1 = a1->mask + (size_t)r1 * L1;
      const uint8_t* m2 = a2->mask + (size_t)r2 * L2;
      for (uint32_t i = 0, ii = 0; i != L1; ++i) {
        if (!m1[i]) continue;
        uint32_t x = m->rowptr[ii];
        const uint32_t xe = m->rowptr[ii + 1];
        for (uint32_t j = 0, jj = 0; j != L2 && x != xe; ++j) {
          if (!m2[j]) continue;
          if (jj == m->col[x]) {
            p[(size_t)i * L2 + j] += m->val[x] / (N1 * N2);
            ++x;
          }
          ++jj;
        }
        ++ii;
      }
    }
  for (size_t e = 0; e < (size_t)L1 * L2; ++e) {
    if (p[e] <= CUTOFF) p[e] = 0.0;
    if (p[e] > 1.0) p[e] = 1.0;
  }
  return p;
}

/* average_basepairing_probability, dafs.cpp:561-607, use_alifold == false branch only */
static float* average_basepairing_probability(const orc_pipeline* pl, const aln_t* a) {
  const uint32_t L = a->L, N = a->n;
  float* p = (float*)calloc((size_t)L * L, sizeof(float));
  uint32_t* idx = (uint32_t*)malloc(((size_t)L + 1) * sizeof(uint32_t));
  for (uint32_t r = 0; r < N; ++r) {
    const uint8_t* m = a->mask + (size_t)r * L;
    for (uint32_t i = 0, j = 0; i != L; ++i)
      if (m[i]) idx[j++] = i;
    const orc_csr* bp = &pl->bp[a->idx[r]];
    for (uint32_t i = 0; i != bp->nrow; ++i)
      for (uint32_t e = bp->rowptr[i]; e < bp->rowptr[i + 1]; ++e)
        p[(size_t)idx[i] * L + idx[bp->col[e]]] += bp->val[e] / N;
  }
  for (uint32_t i = 0; i + 1 < L; ++i)
    for (uint32_t j = i + 1; j != L; ++j)
      if (p[(size_t)i * L + j] <= CUTOFF) p[(size_t)i * L + j] = 0.0;
  free(idx);
  return p;
}

/* DAFS::update_basepairing_probability, dafs.cpp:609-712 (options --bp-update / --bp-update1), use_alifold == false and one
 * level of brackets (th_s_.size() == 1): every sequence of the alignment is folded again under the constraint the
 * decoded common structure ss / str puts on it (paired columns whose two residues exist: '(' and ')'; everything else
 * '?'), and the constrained posteriors are averaged like the unconstrained ones. */
static float* update_basepairing_probability(const orc_pipeline* pl, const aln_t* a, const uint32_t* ss, const char* str) {
  const uint32_t L = a->L, N = a->n;
  float* p = (float*)calloc((size_t)L * L, sizeof(float));
  uint32_t* idx = (uint32_t*)malloc(((size_t)L + 1) * sizeof(uint32_t));
  uint32_t* rev = (uint32_t*)malloc(((size_t)L + 1) * sizeof(uint32_t));
  for (uint32_t r = 0; r < N; ++r) {
    const uint32_t s = a->idx[r], Ls = pl->len[s];
    const uint8_t* m = a->mask + (size_t)r * L;
    for (uint32_t i = 0, j = 0; i != L; ++i) {
      rev[i] = ORC_NONE;
      if (m[i]) { idx[j] = i; rev[i] = j; j++; }
    }
    char* con = (char*)malloc((size_t)Ls + 1);
    memset(con, '?', Ls);
    con[Ls] = 0;
    for (uint32_t i = 0; i != L; ++i)
      if (ss[i] != ORC_NONE && rev[i] != ORC_NONE && rev[ss[i]] != ORC_NONE) {
        if (str[i] == '(') { con[rev[i]] = '('; con[rev[ss[i]]] = ')'; } /* left_brackets[0], fold.cpp:57 */
        else { con[rev[i]] = con[rev[ss[i]]] = '.'; }
      }
    orc_csr bp;
    memset(&bp, 0, sizeof bp);
    csr_alloc(&bp, Ls, (Ls * (Ls + 1)) / 2 + 1);
    const int rc = orc_fold_calculate(pl->seqs[s], Ls, con, (float)CUTOFF, bp.rowptr, bp.col, bp.val); /* s_model_->calculate(seq, con, bp), :659 */
    if (rc >= 0)
      for (uint32_t i = 0; i != Ls; ++i)
        for (uint32_t e = bp.rowptr[i]; e < bp.rowptr[i + 1]; ++e)
          p[(size_t)idx[i] * L + idx[bp.col[e]]] += bp.val[e] / N;
    orc_csr_free(&bp);
    free(con);
  }
  for (uint32_t i = 0; i + 1 < L; ++i)
    for (uint32_t j = i + 1; j != L; ++j)
      if (p[(size_t)i * L + j] <= CUTOFF) p[(size_t)i * L + j] = 0.0;
  free(idx); free(rev);
  return p;
}

/* the `if (use_bp_update_)` blocks of DAFS::align_alignments(VU&, ...), dafs.cpp:919-934: decode, then re-estimate */
static float* bp_update(const orc_pipeline* pl, const aln_t* a, float* p, float th) {
  uint32_t* ss = (uint32_t*)malloc(((size_t)a->L + 1) * sizeof(uint32_t));
  char* str = (char*)malloc((size_t)a->L + 1);
  orc_nussinov_decode(th, 0.0f, a->L, p, NULL, ss);
  orc_make_brackets(a->L, ss, str);
  float* q = update_basepairing_probability(pl, a, ss, str);
  free(ss); free(str); free(p);
  return q;
}

typedef struct { uint32_t i, j, k, l; } cbp_t;
typedef struct { uint32_t* ptr; uint32_t* idx; } adj_t; /* sorted-unique per-row lists (c_x, c_y, c_z) */

static int cmp_u64(const void* a, const void* b) {
  uint64_t x = *(const uint64_t*)a, y = *(const uint64_t*)b;
  return x < y ? -1 : x > y;
}
/* build sorted-unique row lists from (row,col) pairs: dafs.cpp:1046-1060 */
static void adj_build(adj_t* a, uint32_t nrow, uint64_t* keys, size_t nk) {
  qsort(keys, nk, sizeof(uint64_t), cmp_u64);
  size_t u = 0;
  for (size_t e = 0; e < nk; ++e)
    if (e == 0 || keys[e] != keys[e - 1]) keys[u++] = keys[e];
  a->ptr = (uint32_t*)calloc((size_t)nrow + 1, sizeof(uint32_t));
  a->idx = (uint32_t*)malloc((u + 1) * sizeof(uint32_t));
  for (size_t e = 0; e < u; ++e) { a->ptr[(keys[e] >> 32) + 1]++; a->idx[e] = (uint32_t)keys[e]; }
  for (uint32_t r = 0; r < nrow; ++r) a->ptr[r + 1] += a->ptr[r];
}

/* DAFS::solve_by_dd, dafs.cpp:1006-1295 (SPARSE_UPDATE, plain subgradient branch) */
static float solve_by_dd(orc_pipeline* pl, uint32_t* x, uint32_t* y, uint32_t* z,
                         const float* p_x, const float* p_y, const float* p_z,
                         uint32_t L1, uint32_t L2, uint32_t N1, uint32_t N2) {
  const orc_params* prm = &pl->prm;
  const float w_ = prm->w, th_a_ = prm->th_a, eta0_ = prm->eta0;
  const float min_th_s = prm->th_s;
  size_t ncbp = 0, cap = 1024;
  cbp_t* cbp = (cbp_t*)malloc(cap * sizeof(cbp_t));
#define PX(i, j) p_x[(size_t)(i) * L1 + (j)]
#define PY(k, l) p_y[(size_t)(k) * L2 + (l)]
#define PZ(i, k) p_z[(size_t)(i) * L2 + (k)]
  for (uint32_t i = 0; i + 1 < L1; ++i)
    for (uint32_t j = i + 1; j != L1; ++j)
      if (PX(i, j) > CUTOFF)
        for (uint32_t k = 0; k + 1 < L2; ++k)
          if (PZ(i, k) > CUTOFF)
            for (uint32_t l = k + 1; l != L2; ++l)
              if (PY(k, l) > CUTOFF && PZ(j, l) > CUTOFF) {
                float p = (N1 * PX(i, j) + N2 * PY(k, l)) / (N1 + N2);
                float q = (PZ(i, k) + PZ(j, l)) / 2;
                if (p - min_th_s > 0.0 && w_ * (p - min_th_s) + (q - th_a_) > 0.0) {
                  if (ncbp == cap) { cap *= 2; cbp = (cbp_t*)realloc(cbp, cap * sizeof(cbp_t)); }
                  cbp_t c = {i, j, k, l};
                  cbp[ncbp++] = c;
                }
              }
  adj_t c_x, c_y, c_z;
  {
    uint64_t* kx = (uint64_t*)malloc((ncbp + 1) * sizeof(uint64_t));
    uint64_t* ky = (uint64_t*)malloc((ncbp + 1) * sizeof(uint64_t));
    uint64_t* kz = (uint64_t*)malloc((2 * ncbp + 1) * sizeof(uint64_t));
    for (size_t u = 0; u < ncbp; ++u) {
      kx[u] = ((uint64_t)cbp[u].i << 32) | cbp[u].j;
      ky[u] = ((uint64_t)cbp[u].k << 32) | cbp[u].l;
      kz[2 * u] = ((uint64_t)cbp[u].i << 32) | cbp[u].k;
      kz[2 * u + 1] = ((uint64_t)cbp[u].j << 32) | cbp[u].l;
    }
    adj_build(&c_x, L1, kx, ncbp);
    adj_build(&c_y, L2, ky, ncbp);
    adj_build(&c_z, L1, kz, 2 * ncbp);
    free(kx); free(ky); free(kz);
  }

  uint32_t* env = (uint32_t*)malloc(2 * ((size_t)L1 + 1) * sizeof(uint32_t));
  orc_nw_envelope(th_a_, L1, L2, p_z, env); /* a_decoder_->initialize(p_z), :1064 */

  float* q_x = (float*)calloc((size_t)L1 * L1, sizeof(float));
  float* q_y = (float*)calloc((size_t)L2 * L2, sizeof(float));
  float* q_z = (float*)calloc((size_t)L1 * L2, sizeof(float));
  int* t_x = (int*)malloc((size_t)L1 * L1 * sizeof(int));
  int* t_y = (int*)malloc((size_t)L2 * L2 * sizeof(int));
  int* t_z = (int*)malloc((size_t)L1 * L2 * sizeof(int));
#define QX(i, j) q_x[(size_t)(i) * L1 + (j)]
#define QY(k, l) q_y[(size_t)(k) * L2 + (l)]
#define QZ(i, k) q_z[(size_t)(i) * L2 + (k)]
#define TX(i, j) t_x[(size_t)(i) * L1 + (j)]
#define TY(k, l) t_y[(size_t)(k) * L2 + (l)]
#define TZ(i, k) t_z[(size_t)(i) * L2 + (k)]
  float c = 0.0;
  float eta = eta0_;
  float s_prev = 0.0;
  uint32_t violated = 0;
  uint32_t t;
  for (t = 0; t != prm->t_max; ++t) {
    float s = 0.0;
    s += orc_nussinov_decode(prm->th_s, w_ * 2 * N1 / (N1 + N2), L1, p_x, q_x, x);
    s += orc_nussinov_decode(prm->th_s, w_ * 2 * N2 / (N1 + N2), L2, p_y, q_y, y);
    s += orc_nw_decode(th_a_, L1, L2, p_z, q_z, env, z);

    violated = 0;
    memset(t_x, 0, (size_t)L1 * L1 * sizeof(int));
    memset(t_y, 0, (size_t)L2 * L2 * sizeof(int));
    memset(t_z, 0, (size_t)L1 * L2 * sizeof(int));
    for (size_t u = 0; u != ncbp; ++u) {
      const uint32_t i = cbp[u].i, j = cbp[u].j, k = cbp[u].k, l = cbp[u].l;
      const float s_w = QX(i, j) + QY(k, l) - QZ(i, k) - QZ(j, l);
      if (s_w > 0.0f) {
        s += s_w;
        TX(i, j)++; TY(k, l)++; TZ(i, k)++; TZ(j, l)++;
      }
    }
    for (uint32_t i = 0; i != L1; ++i) { /* :1121-1150 */
      const uint32_t j = x[i];
      if (j != ORC_NONE && TX(i, j) != 1) { violated++; QX(i, j) -= eta * (TX(i, j) - 1); }
      for (uint32_t e = c_x.ptr[i]; e != c_x.ptr[i + 1]; ++e) {
        const uint32_t jj = c_x.idx[e];
        if (x[i] != jj && TX(i, jj) != 0) { violated++; QX(i, jj) -= eta * TX(i, jj); }
      }
    }
    for (uint32_t k = 0; k != L2; ++k) { /* :1172-1201 */
      const uint32_t l = y[k];
      if (l != ORC_NONE && TY(k, l) != 1) { violated++; QY(k, l) -= eta * (TY(k, l) - 1); }
      for (uint32_t e = c_y.ptr[k]; e != c_y.ptr[k + 1]; ++e) {
        const uint32_t ll = c_y.idx[e];
        if (y[k] != ll && TY(k, ll) != 0) { violated++; QY(k, ll) -= eta * TY(k, ll); }
      }
    }
    for (uint32_t i = 0; i != L1; ++i) { /* :1223-1254; std::max(a,b) = (a<b)?b:a */
      const uint32_t k = z[i];
      if (k != ORC_NONE) {
        if (TZ(i, k) > 1) violated++;
        float v = QZ(i, k) - eta * (1 - TZ(i, k));
        QZ(i, k) = (0.0f < v) ? v : 0.0f;
      }
      for (uint32_t e = c_z.ptr[i]; e != c_z.ptr[i + 1]; ++e) {
        const uint32_t kk = c_z.idx[e];
        if (z[i] != kk) {
          if (TZ(i, kk) > 0) violated++;
          float v = QZ(i, kk) + eta * TZ(i, kk);
          QZ(i, kk) = (0.0f < v) ? v : 0.0f;
        }
      }
    }
    if (violated == 0 && !prm->force_iters) break;
    if (s > s_prev || t == 0) { /* :1283-1288 */
      float num = 4.0f * ncbp - violated;
      num = (0.0f < num) ? num : 0.0f;
      c += num / (4.0 * ncbp);
      eta = eta0_ / (1.0 + c);
    }
    s_prev = s;
  }
  pl->dd_iters[pl->dd_n] = t;
  pl->dd_viol[pl->dd_n] = violated;
  pl->dd_n++;
  free(cbp); free(c_x.ptr); free(c_x.idx); free(c_y.ptr); free(c_y.idx); free(c_z.ptr); free(c_z.idx);
  free(env); free(q_x); free(q_y); free(q_z); free(t_x); free(t_y); free(t_z);
  return s_prev;
}

/* project_alignment, dafs.cpp:766-825 */
static void project_alignment(aln_t* out, const aln_t* a1, const aln_t* a2, const uint32_t* z) {
  const uint32_t L1 = a1->L, L2 = a2->L;
  uint32_t c = 0;
  for (uint32_t i = 0; i != L1; ++i)
    if (z[i] != ORC_NONE) c++;
  const uint32_t L = L1 + L2 - c;
  out->n = a1->n + a2->n;
  out->L = L;
  out->idx = (uint32_t*)malloc(out->n * sizeof(uint32_t));
  out->mask = (uint8_t*)calloc((size_t)out->n * L, 1);
  uint32_t row = 0;
  for (uint32_t q = 0; q < a1->n; ++q, ++row) {
    out->idx[row] = a1->idx[q];
    uint8_t* p = out->mask + (size_t)row * L;
    const uint8_t* s = a1->mask + (size_t)q * L1;
    uint32_t r = 0, k = 0;
    for (uint32_t i = 0; i != L1; ++i) {
      if (z[i] != ORC_NONE) {
        while (k < z[i]) { p[r++] = 0; k++; }
        p[r++] = s[i];
        ++k;
      } else p[r++] = s[i];
    }
    while (k < L2) { p[r++] = 0; k++; }
  }
  for (uint32_t q = 0; q < a2->n; ++q, ++row) {
    out->idx[row] = a2->idx[q];
    uint8_t* p = out->mask + (size_t)row * L;
    const uint8_t* s = a2->mask + (size_t)q * L2;
    uint32_t k = 0, r = 0;
    for (uint32_t i = 0; i != L1; ++i) {
      if (z[i] != ORC_NONE) {
        while (k < z[i]) p[r++] = s[k++];
        p[r++] = s[k++];
      } else p[r++] = 0;
    }
    while (k < L2) p[r++] = s[k++];
  }
}

/* DAFS::align(ALN&, int), dafs.cpp:1499-1516 + align_alignments :896-911 */
static void align_node(orc_pipeline* pl, aln_t* out, uint32_t ch) {
  if (pl->tleft[ch] == ORC_NONE) {
    out->n = 1;
    out->L = pl->len[ch];
    out->idx = (uint32_t*)malloc(sizeof(uint32_t));
    out->idx[0] = ch;
    out->mask = (uint8_t*)malloc(out->L ? out->L : 1);
    memset(out->mask, 1, out->L);
    return;
  }
  aln_t a1, a2;
  align_node(pl, &a1, pl->tleft[ch]);
  align_node(pl, &a2, pl->tright[ch]);
  float* p_x = average_basepairing_probability(pl, &a1);
  float* p_y = average_basepairing_probability(pl, &a2);
  if (pl->prm.bp_update && ch == 2 * pl->N - 2) { /* only the top call takes the overload with the update, dafs.cpp:1518-1537 */
    p_x = bp_update(pl, &a1, p_x, pl->prm.th_s);
    p_y = bp_update(pl, &a2, p_y, pl->prm.th_s);
  }
  float* p_z = average_matching_probability(pl, &a1, &a2);
  uint32_t* x = (uint32_t*)malloc(((size_t)a1.L + 1) * sizeof(uint32_t));
  uint32_t* y = (uint32_t*)malloc(((size_t)a2.L + 1) * sizeof(uint32_t));
  uint32_t* z = (uint32_t*)malloc(((size_t)a1.L + 1) * sizeof(uint32_t));
  solve_by_dd(pl, x, y, z, p_x, p_y, p_z, a1.L, a2.L, a1.n, a2.n);
  project_alignment(out, &a1, &a2, z);
  free(p_x); free(p_y); free(p_z); free(x); free(y); free(z);
  aln_free(&a1); aln_free(&a2);
}

/* ---------------- pipeline object ---------------- */
orc_pipeline* orc_pipeline_new(const orc_params* prm, uint32_t N, const char* const* names, const char* const* seqs) {
  orc_pipeline* pl = (orc_pipeline*)calloc(1, sizeof(orc_pipeline));
  pl->prm = *prm;
  pl->N = N;
  pl->names = (char**)malloc(N * sizeof(char*));
  pl->seqs = (char**)malloc(N * sizeof(char*));
  pl->len = (uint32_t*)malloc(N * sizeof(uint32_t));
  for (uint32_t i = 0; i < N; ++i) {
    pl->names[i] = strdup(names[i]);
    pl->seqs[i] = strdup(seqs[i]);
    pl->len[i] = (uint32_t)strlen(seqs[i]);
  }
  pl->bp = (orc_csr*)calloc(N, sizeof(orc_csr));
  pl->mp = (orc_csr*)calloc((size_t)N * N, sizeof(orc_csr));
  pl->sim = (float*)calloc((size_t)N * N, sizeof(float));
  pl->tscore = (float*)calloc(2 * N, sizeof(float));
  pl->tleft = (uint32_t*)calloc(2 * N, sizeof(uint32_t));
  pl->tright = (uint32_t*)calloc(2 * N, sizeof(uint32_t));
  pl->dd_iters = (uint32_t*)calloc(N + 1, sizeof(uint32_t));
  pl->dd_viol = (uint32_t*)calloc(N + 1, sizeof(uint32_t));
  return pl;
}

void orc_pipeline_free(orc_pipeline* pl) {
  if (!pl) return;
  for (uint32_t i = 0; i < pl->N; ++i) { free(pl->names[i]); free(pl->seqs[i]); orc_csr_free(&pl->bp[i]); }
  for (size_t e = 0; e < (size_t)pl->N * pl->N; ++e) orc_csr_free(&pl->mp[e]);
  free(pl->names); free(pl->seqs); free(pl->len); free(pl->bp); free(pl->mp); free(pl->sim);
  free(pl->tscore); free(pl->tleft); free(pl->tright); free(pl->dd_iters); free(pl->dd_viol);
  aln_free(&pl->final_aln); free(pl->final_ss); free(pl->final_str); free(pl->out);
  free(pl);
}

void orc_pipeline_set_bp(orc_pipeline* pl, uint32_t x, const uint32_t* rowptr, const uint32_t* col, const float* val) {
  const uint32_t L = pl->len[x];
  orc_csr_free(&pl->bp[x]);
  csr_alloc(&pl->bp[x], L, rowptr[L]);
  memcpy(pl->bp[x].rowptr, rowptr, ((size_t)L + 1) * sizeof(uint32_t));
  memcpy(pl->bp[x].col, col, (size_t)rowptr[L] * sizeof(uint32_t));
  memcpy(pl->bp[x].val, val, (size_t)rowptr[L] * sizeof(float));
}

/* --align-aux equivalent (AUXAlign, src/align.cpp:190-247): rows of mp[x][y], x < y, supplied by the caller; used with
 * prm.align_model == 2.  The transpose is built in phase 1 as for a computed matrix. */
void orc_pipeline_set_mp(orc_pipeline* pl, uint32_t x, uint32_t y, const uint32_t* rowptr, const uint32_t* col, const float* val) {
  const uint32_t L = pl->len[x];
  orc_csr* m = MP(pl, x, y);
  orc_csr_free(m);
  csr_alloc(m, L, rowptr[L]);
  memcpy(m->rowptr, rowptr, ((size_t)L + 1) * sizeof(uint32_t));
  memcpy(m->col, col, (size_t)rowptr[L] * sizeof(uint32_t));
  memcpy(m->val, val, (size_t)rowptr[L] * sizeof(float));
}

/* DAFS::run, dafs.cpp:1787-1830 */
int orc_pipeline_phase1(orc_pipeline* pl) {
  const uint32_t N = pl->N;
  double t0 = now_s();
  /* s_model_->calculate(fa_, bp_) :1787; CONTRAfold(CUTOFF) :1704 */
  if (pl->prm.fold_model == 0) {
    for (uint32_t x = 0; x < N; ++x) {
      const uint32_t L = pl->len[x];
      orc_csr_free(&pl->bp[x]);
      csr_alloc(&pl->bp[x], L, (L * (L + 1)) / 2 + 1);
      int rc = orc_fold_calculate(pl->seqs[x], L, NULL, (float)CUTOFF, pl->bp[x].rowptr, pl->bp[x].col, pl->bp[x].val);
      if (rc < 0) return rc;
    }
  } else {
    for (uint32_t x = 0; x < N; ++x)
      if (!pl->bp[x].rowptr) return -2;
  }
  double t1 = now_s();
  pl->secs[0] = t1 - t0;
  /* a_model_->calculate(fa_, mp_) :1796 (align.cpp:35-52) + transposes :1797-1799 */
  for (uint32_t i = 0; i < N; ++i) {
    orc_csr_free(MP(pl, i, i));
    csr_identity(MP(pl, i, i), pl->len[i]);
    for (uint32_t j = i + 1; j < N; ++j) {
      const uint32_t L1 = pl->len[i], L2 = pl->len[j];
      orc_csr* m = MP(pl, i, j);
      if (pl->prm.align_model == 2) { /* rows supplied through orc_pipeline_set_mp */
        if (!m->rowptr || m->nrow != L1) return -3;
      } else {
        orc_csr_free(m);
        csr_alloc(m, L1, L1 * L2);
        int rc = orc_align_calculate(pl->prm.align_model, pl->seqs[i], L1, pl->seqs[j], L2, pl->prm.th_a, m->rowptr, m->col, m->val);
        if (rc < 0) return rc;
        m->col = (uint32_t*)realloc(m->col, ((size_t)rc + 1) * sizeof(uint32_t));
        m->val = (float*)realloc(m->val, ((size_t)rc + 1) * sizeof(float));
      }
      orc_csr_free(MP(pl, j, i));
      orc_transpose(m, L2, MP(pl, j, i));
    }
  }
  double t2 = now_s();
  pl->secs[1] = t2 - t1;
  if (pl->prm.w_pct_f != 0.0) relax_fourway_consistency(pl); /* :1808-1809, before the similarity scores */
  /* sim_ :1813-1819 */
  for (uint32_t i = 0; i < N; ++i) {
    SIM(pl, i, i) = 1.0;
    for (uint32_t j = i + 1; j < N; ++j) {
      const orc_csr* m = MP(pl, i, j);
      SIM(pl, i, j) = SIM(pl, j, i) = orc_similarity_score(m->rowptr, m->col, m->val, pl->len[i], pl->len[j]);
    }
  }
  if (pl->prm.w_pct_s != 0.0) relax_basepairing_probability(pl); /* :1822-1823 */
  if (pl->prm.w_pct_a != 0.0) relax_matching_probability(pl);    /* :1826-1827 */
  build_tree(pl);                                                /* :1830 */
  pl->out_len = 0;
  print_tree(pl, 2 * N - 2);
  out_str(pl, "\n");
  pl->secs[2] = now_s() - t2;
  return 0;
}

static int cmp_row(const void* a, const void* b) {
  uint32_t x = *(const uint32_t*)a, y = *(const uint32_t*)b;
  return x < y ? -1 : x > y;
}

/* DAFS::run, dafs.cpp:1835-1879 (n_refinement_ = 0, no bp-update) */
int orc_pipeline_phase2(orc_pipeline* pl) {
  const uint32_t N = pl->N;
  double t0 = now_s();
  pl->dd_n = 0;
  aln_free(&pl->final_aln);
  align_node(pl, &pl->final_aln, 2 * N - 2);
  aln_t* a = &pl->final_aln;
  /* final common structure :1857-1871 -- WITHOUT the alifold term (see header) */
  float* p = average_basepairing_probability(pl, a);
  if (pl->prm.bp_update1) p = bp_update(pl, a, p, pl->prm.th_s1); /* :1863-1869 */
  free(pl->final_ss); free(pl->final_str);
  pl->final_ss = (uint32_t*)malloc(((size_t)a->L + 1) * sizeof(uint32_t));
  pl->final_str = (char*)malloc((size_t)a->L + 1);
  orc_nussinov_decode(pl->prm.th_s1, 0.0f, a->L, p, NULL, pl->final_ss);
  orc_make_brackets(a->L, pl->final_ss, pl->final_str);
  free(p);
  /* std::sort(aln) :1876 sorts by (seq index, mask); indices are unique */
  uint32_t* order = (uint32_t*)malloc(2 * (size_t)a->n * sizeof(uint32_t));
  for (uint32_t r = 0; r < a->n; ++r) { order[2 * r] = a->idx[r]; order[2 * r + 1] = r; }
  qsort(order, a->n, 2 * sizeof(uint32_t), cmp_row);
  out_str(pl, ">SS_cons\n");
  out_str(pl, pl->final_str);
  out_str(pl, "\n");
  char* line = (char*)malloc((size_t)a->L + 2);
  for (uint32_t o = 0; o < a->n; ++o) { /* output, :1584-1601 */
    const uint32_t r = order[2 * o + 1], s = a->idx[r];
    out_str(pl, "> ");
    out_str(pl, pl->names[s]);
    out_str(pl, "\n");
    const uint8_t* m = a->mask + (size_t)r * a->L;
    for (uint32_t j = 0, k = 0; j != a->L; ++j) line[j] = m[j] ? pl->seqs[s][k++] : '-';
    line[a->L] = '\n';
    out_put(pl, line, (size_t)a->L + 1);
  }
  free(line); free(order);
  pl->secs[3] = now_s() - t0;
  return 0;
}

const orc_csr* orc_pipeline_mp(const orc_pipeline* pl, uint32_t x, uint32_t y) { return MP(pl, x, y); }
const orc_csr* orc_pipeline_bp(const orc_pipeline* pl, uint32_t x) { return &pl->bp[x]; }
const float* orc_pipeline_sim(const orc_pipeline* pl) { return pl->sim; }
void orc_pipeline_tree(const orc_pipeline* pl, float* score, uint32_t* left, uint32_t* right) {
  const uint32_t T = 2 * pl->N - 1;
  memcpy(score, pl->tscore, T * sizeof(float));
  memcpy(left, pl->tleft, T * sizeof(uint32_t));
  memcpy(right, pl->tright, T * sizeof(uint32_t));
}
const char* orc_pipeline_output(orc_pipeline* pl) { return pl->out ? pl->out : ""; }
uint32_t orc_pipeline_dd_log(const orc_pipeline* pl, uint32_t* iters, uint32_t* violated, uint32_t cap) {
  uint32_t n = pl->dd_n < cap ? pl->dd_n : cap;
  memcpy(iters, pl->dd_iters, n * sizeof(uint32_t));
  memcpy(violated, pl->dd_viol, n * sizeof(uint32_t));
  return pl->dd_n;
}
double orc_pipeline_seconds(const orc_pipeline* pl, int phase) { return (phase >= 0 && phase < 4) ? pl->secs[phase] : 0.0; }
