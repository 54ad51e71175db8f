/* oracle/decoders.c -- TEST INFRASTRUCTURE (see oracle.h).
 *
 * CPU restatement of the two per-iteration DP decoders DAFS instantiates
 * (src/dafs.cpp:1692,1759-1760): SparseNussinov and SparseNeedlemanWunsch.
 * PINNED: bit-exact (traceback arrays and scores) against oracle/_ref on the committed
 * fixtures tests/golden/decoders_*.npz.
 * Also the dense classes Nussinov / NeedlemanWunsch (never instantiated by DAFS), pinned the same way
 * on tests/golden/decoders_dense.npz.
 */
#include "oracle.h"
#include <float.h>
#include <stdlib.h>
#include <string.h>

/* SparseNussinov::decode, src/nussinov.cpp:207-298 (with q) and :300-392 (q==NULL: s = p-th).
 * Candidate lists bp[j] are kept per column in insertion order (= decreasing k). */
float orc_nussinov_decode(float th, float w, uint32_t L, const float* p, const float* q, uint32_t* ss) {
  for (uint32_t i = 0; i < L; ++i) ss[i] = ORC_NONE;
  if (L == 0) return 0.0f;
  float* dp = (float*)calloc((size_t)L * L, sizeof(float));
  uint32_t* tr = (uint32_t*)calloc((size_t)L * L, sizeof(uint32_t));
  /* column lists: at most L entries per column */
  uint32_t* cnt = (uint32_t*)calloc(L, sizeof(uint32_t));
  uint32_t* ck = (uint32_t*)malloc((size_t)L * L * sizeof(uint32_t));
  float* cs = (float*)malloc((size_t)L * L * sizeof(float));
#define DP(i, j) dp[(size_t)(i) * L + (j)]
  for (uint32_t l = 1; l < L; ++l) {
    for (uint32_t i = 0; i + l < L; ++i) {
      uint32_t j = i + l;
      float v = 0.0f;
      int t = 0;
      if (i + 1 < j) { v = DP(i + 1, j); t = 1; }
      if (i < j - 1 && v < DP(i, j - 1)) { v = DP(i, j - 1); t = 2; }
      if (i + 1 < j - 1) {
        float s = q ? w * (p[(size_t)i * L + j] - th) - q[(size_t)i * L + j] : p[(size_t)i * L + j] - th;
        if (s > 0.0) {
          ck[(size_t)j * L + cnt[j]] = i;
          cs[(size_t)j * L + cnt[j]] = DP(i + 1, j - 1) + s;
          cnt[j]++;
          if (v < DP(i + 1, j - 1) + s) { v = DP(i + 1, j - 1) + s; t = 3; }
        }
      }
      for (uint32_t x = 0; x < cnt[j]; ++x) {
        const uint32_t k = ck[(size_t)j * L + x];
        const float s = cs[(size_t)j * L + x];
        if (i < k) {
          if (v < DP(i, k - 1) + s) { v = DP(i, k - 1) + s; t = (int)(k - i + 3); }
        }
      }
      DP(i, j) = v;
      tr[(size_t)i * L + j] = (uint32_t)t;
    }
  }
  /* traceback, :265-295 (explicit stack) */
  uint32_t* st = (uint32_t*)malloc((size_t)(2 * L + 4) * 2 * sizeof(uint32_t));
  size_t sp = 0;
  st[0] = 0; st[1] = L - 1; sp = 1;
  while (sp) {
    --sp;
    const int i = (int)st[2 * sp], j = (int)st[2 * sp + 1];
    uint32_t t = tr[(size_t)i * L + j];
    switch (t) {
      case 0: break;
      case 1: st[2 * sp] = i + 1; st[2 * sp + 1] = j; ++sp; break;
      case 2: st[2 * sp] = i; st[2 * sp + 1] = j - 1; ++sp; break;
      case 3: ss[i] = j; st[2 * sp] = i + 1; st[2 * sp + 1] = j - 1; ++sp; break;
      default: {
        const int k = i + (int)t - 3;
        st[2 * sp] = i; st[2 * sp + 1] = k - 1; ++sp;
        ss[k] = j;
        st[2 * sp] = k + 1; st[2 * sp + 1] = j - 1; ++sp;
      } break;
    }
  }
  float r = DP(0, L - 1);
#undef DP
  free(st); free(dp); free(tr); free(cnt); free(ck); free(cs);
  return r;
}

/* make_brackets, src/nussinov.cpp:401-413 with left/right_brackets[0] = '(' ')' (fold.cpp:57-58) */
void orc_make_brackets(uint32_t L, const uint32_t* ss, char* str) {
  memset(str, '.', L);
  str[L] = 0;
  for (uint32_t i = 0; i != L; ++i)
    if (ss[i] != ORC_NONE) { str[i] = '('; str[ss[i]] = ')'; }
}

/* SparseNeedlemanWunsch::initialize, src/needleman_wunsch.cpp:198-253 */
void orc_nw_envelope(float th, uint32_t L1, uint32_t L2, const float* p, uint32_t* env) {
#define FIRST(i) env[2 * (i)]
#define SECOND(i) env[2 * (i) + 1]
  for (uint32_t i = 0; i <= L1; ++i) { FIRST(i) = 0; SECOND(i) = 0; }
  for (uint32_t i = 1; i != L1 + 1; ++i) {
    for (uint32_t k = 1; k != L2 + 1; ++k) {
      if (p[(size_t)(i - 1) * L2 + (k - 1)] - th >= 0.0) {
        if (k - 1 < FIRST(i - 1)) FIRST(i - 1) = k - 1;
        FIRST(i) = k;
        break;
      }
    }
    if (FIRST(i) == 0) {
      FIRST(i) = FIRST(i - 1);
      SECOND(i) = SECOND(i - 1);
      continue;
    }
    for (uint32_t k = L2; k != 0; --k) {
      if (p[(size_t)(i - 1) * L2 + (k - 1)] - th >= 0.0) {
        if (k - 1 > SECOND(i - 1)) SECOND(i - 1) = k - 1;
        SECOND(i) = k;
        break;
      }
    }
  }
  SECOND(L1) = L2;
  for (uint32_t i = L1, v = L2; i != 0; --i) { v = v < FIRST(i) ? v : FIRST(i); FIRST(i) = v; }
  for (uint32_t i = 0, v = 0; i != L1 + 1; ++i) { v = v > SECOND(i) ? v : SECOND(i); SECOND(i) = v; }
  for (uint32_t i = 1; i != L1 + 1; ++i)
    if (SECOND(i - 1) < FIRST(i)) FIRST(i) = SECOND(i - 1);
}

/* SparseNeedlemanWunsch::decode, :255-338 (q) / :340-422 (q==NULL) */
float orc_nw_decode(float th, uint32_t L1, uint32_t L2, const float* p, const float* q,
                    const uint32_t* env, uint32_t* al) {
  const size_t W = (size_t)L2 + 1;
  float* dp = (float*)malloc((size_t)(L1 + 1) * W * sizeof(float));
  char* tr = (char*)malloc((size_t)(L1 + 1) * W);
  for (size_t c = 0; c < (size_t)(L1 + 1) * W; ++c) { dp[c] = -FLT_MAX; tr[c] = ' '; }
  dp[0] = 0.0f;
  for (uint32_t i = 1; i != L1 + 1; ++i) { dp[i * W] = 0.0f; tr[i * W] = 'X'; }
  for (uint32_t k = 1; k != L2 + 1; ++k) { dp[k] = 0.0f; tr[k] = 'Y'; }
  for (uint32_t i = 1; i != L1 + 1; ++i) {
    for (uint32_t k = FIRST(i); k <= SECOND(i); ++k) {
      if (k == 0) continue;
      float v = dp[(i - 1) * W + (k - 1)] + p[(size_t)(i - 1) * L2 + (k - 1)] - th;
      if (q) v = v + q[(size_t)(i - 1) * L2 + (k - 1)];
      char t = 'M';
      if (v < dp[(i - 1) * W + k]) { v = dp[(i - 1) * W + k]; t = 'X'; }
      if (v < dp[i * W + (k - 1)]) { v = dp[i * W + (k - 1)]; t = 'Y'; }
      dp[i * W + k] = v;
      tr[i * W + k] = t;
    }
  }
#undef FIRST
#undef SECOND
  /* traceback :298-335.  The reference asserts the path never leaves the envelope; a
   * cell with tr==' ' would loop forever there, so it is reported as a failed decode (NaN). */
  char* rpath = (char*)malloc((size_t)L1 + L2 + 2);
  size_t n = 0;
  int i = (int)L1, k = (int)L2, bad = 0;
  while (i > 0 || k > 0) {
    char t = tr[i * W + k];
    rpath[n++] = t;
    if (t == 'M') { --i; --k; }
    else if (t == 'X') --i;
    else if (t == 'Y') --k;
    else { bad = 1; break; }
  }
  for (uint32_t a = 0; a < L1; ++a) al[a] = ORC_NONE;
  uint32_t ai = 0, ak = 0;
  for (size_t x = n; x-- > 0;) {
    switch (rpath[x]) {
      case 'M': al[ai++] = ak++; break;
      case 'X': al[ai++] = ORC_NONE; break;
      case 'Y': ak++; break;
      default: break;
    }
  }
  float r = dp[(size_t)L1 * W + L2];
  free(rpath); free(dp); free(tr);
  if (bad) return 0.0f / 0.0f;
  return r;
}


/* ------------------------------------------------------------------------------------------------------------
 * Dense decoders: Nussinov::decode, src/nussinov.cpp:32-113 (with q) and :115-204 (q == NULL:
 * s = p - th): every pair scores (no "positive only" filter), the bifurcation runs over every split k.
 * ---------------------------------------------------------------------------------------------------------- */
float orc_nussinov_dense_decode(float th, float w, uint32_t L, const float* p, const float* q, uint32_t* ss) {
  for (uint32_t i = 0; i < L; ++i) ss[i] = ORC_NONE;
  if (L == 0) return 0.0f;
  float* dp = (float*)calloc((size_t)L * L, sizeof(float));
  uint32_t* tr = (uint32_t*)calloc((size_t)L * L, sizeof(uint32_t));
#define DD(i, j) dp[(size_t)(i) * L + (j)]
  for (uint32_t l = 1; l < L; ++l)
    for (uint32_t i = 0; i + l < L; ++i) {
      const uint32_t j = i + l;
      float v = 0.0f;
      int t = 0;
      if (i + 1 < j) { v = DD(i + 1, j); t = 1; }
      if (i < j - 1 && v < DD(i, j - 1)) { v = DD(i, j - 1); t = 2; }
      {
        const float sm = q ? w * (p[(size_t)i * L + j] - th) - q[(size_t)i * L + j] : p[(size_t)i * L + j] - th;
        if (i + 1 < j - 1 && v < DD(i + 1, j - 1) + sm) { v = DD(i + 1, j - 1) + sm; t = 3; }
      }
      for (uint32_t k = i + 1; k < j; ++k)
        if (v < DD(i, k) + DD(k + 1, j)) { v = DD(i, k) + DD(k + 1, j); t = (int)(k - i + 3); }
      DD(i, j) = v;
      tr[(size_t)i * L + j] = (uint32_t)t;
    }
  uint32_t* st = (uint32_t*)malloc((size_t)(2 * L + 4) * 2 * sizeof(uint32_t));
  size_t sp = 0;
  st[0] = 0; st[1] = L - 1; sp = 1;
  while (sp) {
    --sp;
    const int i = (int)st[2 * sp], j = (int)st[2 * sp + 1];
    const uint32_t t = tr[(size_t)i * L + j];
    if (t == 0) continue;
    if (t == 1) { st[2 * sp] = i + 1; st[2 * sp + 1] = j; ++sp; }
    else if (t == 2) { st[2 * sp] = i; st[2 * sp + 1] = j - 1; ++sp; }
    else if (t == 3) { ss[i] = j; st[2 * sp] = i + 1; st[2 * sp + 1] = j - 1; ++sp; }
    else {
      const int k = i + (int)t - 3;
      st[2 * sp] = i; st[2 * sp + 1] = k; ++sp;
      st[2 * sp] = k + 1; st[2 * sp + 1] = j; ++sp;
    }
  }
  const float r = DD(0, L - 1);
#undef DD
  free(st); free(tr); free(dp);
  return r;
}

/* NeedlemanWunsch::decode, src/needleman_wunsch.cpp:28-196: the whole grid, no envelope */
float orc_nw_dense_decode(float th, uint32_t L1, uint32_t L2, const float* p, const float* q, uint32_t* al) {
  const size_t W = (size_t)L2 + 1;
  float* dp = (float*)malloc((size_t)(L1 + 1) * W * sizeof(float));
  char* tr = (char*)malloc((size_t)(L1 + 1) * W);
  for (size_t c = 0; c < (size_t)(L1 + 1) * W; ++c) { dp[c] = -FLT_MAX; tr[c] = ' '; }
  dp[0] = 0.0f;
  for (uint32_t i = 1; i <= L1; ++i) { dp[(size_t)i * W] = 0.0f; tr[(size_t)i * W] = 'X'; }
  for (uint32_t k = 1; k <= L2; ++k) { dp[k] = 0.0f; tr[k] = 'Y'; }
  for (uint32_t i = 1; i <= L1; ++i)
    for (uint32_t k = 1; k <= L2; ++k) {
      float v = dp[(size_t)(i - 1) * W + (k - 1)] + p[(size_t)(i - 1) * L2 + (k - 1)] - th;
      if (q) v = v + q[(size_t)(i - 1) * L2 + (k - 1)];
      char t = 'M';
      if (v < dp[(size_t)(i - 1) * W + k]) { v = dp[(size_t)(i - 1) * W + k]; t = 'X'; }
      if (v < dp[(size_t)i * W + (k - 1)]) { v = dp[(size_t)i * W + (k - 1)]; t = 'Y'; }
      dp[(size_t)i * W + k] = v;
      tr[(size_t)i * W + k] = t;
    }
  for (uint32_t i = 0; i < L1; ++i) al[i] = ORC_NONE;
  int i = (int)L1, k = (int)L2;
  while (i > 0 || k > 0) {
    const char t = tr[(size_t)i * W + k];
    if (t == 'M') { al[i - 1] = (uint32_t)(k - 1); --i; --k; }
    else if (t == 'X') { al[i - 1] = ORC_NONE; --i; }
    else --k;
  }
  const float r = dp[(size_t)L1 * W + L2];
  free(tr); free(dp);
  return r;
}
