/* oracle/probcons.c -- TEST INFRASTRUCTURE (see oracle.h).
 *
 * CPU restatement of the ProbCons 3-state pair-HMM posterior used by DAFS
 * (reference src/probconsRNA, built with -DNumInsertStates=1).
 * PINNED: bit-exact against oracle/_ref (the reference's own sources compiled here) on the
 * committed fixtures tests/golden/probcons_*.npz, and through them against README.md:59.
 */
#include "oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define LOG_ZERO (-2e20f)             /* ScoreType.h:18 */
#define LOG_UNDERFLOW_THRESHOLD 7.5f  /* ScoreType.h:179 */

/* ScoreType.h:187-198 -- float literals, Horner form, no contraction */
static inline float LOOKUP(float x) {
  if (x <= 1.00f) return ((-0.009350833524763f * x + 0.130659527668286f) * x + 0.498799810682272f) * x + 0.693203116424741f;
  if (x <= 2.50f) return ((-0.014532321752540f * x + 0.139942324101744f) * x + 0.495635523139337f) * x + 0.692140569840976f;
  if (x <= 4.50f) return ((-0.004605031767994f * x + 0.063427417320019f) * x + 0.695956496475118f) * x + 0.514272634594009f;
  return ((-0.000458661602210f * x + 0.009695946122598f) * x + 0.930734667215156f) * x + 0.168037164329057f;
}
/* ScoreType.h:233-238 */
static inline void LOG_PLUS_EQUALS(float* x, float y) {
  if (*x < y) *x = (*x == LOG_ZERO || y - *x >= LOG_UNDERFLOW_THRESHOLD) ? y : LOOKUP(y - *x) + *x;
  else        *x = (y == LOG_ZERO || *x - y >= LOG_UNDERFLOW_THRESHOLD) ? *x : LOOKUP(*x - y) + y;
}
/* ScoreType.h:259-262 */
static inline float LOG_ADD(float x, float y) {
  if (x < y) return (x == LOG_ZERO || y - x >= LOG_UNDERFLOW_THRESHOLD) ? y : LOOKUP(y - x) + x;
  return (y == LOG_ZERO || x - y >= LOG_UNDERFLOW_THRESHOLD) ? x : LOOKUP(x - y) + y;
}
/* ScoreType.h:37-57 -- the polynomial is evaluated in double (double literals), result narrowed to float */
static inline float EXP(float x) {
  if (x > -2) {
    if (x > -0.5) {
      if (x > 0) return (float)exp((double)x); /* unreachable here: argument is min(0,.) */
      return (float)((((0.03254409303190190000 * x + 0.16280432765779600000) * x + 0.49929760485974900000) * x + 0.99995149601363700000) * x + 0.99999925508501600000);
    }
    if (x > -1)
      return (float)((((0.01973899026052090000 * x + 0.13822379685007000000) * x + 0.48056651562365000000) * x + 0.99326940370383500000) * x + 0.99906756856399500000);
    return (float)((((0.00940528203591384000 * x + 0.09414963667859410000) * x + 0.40825793595877300000) * x + 0.93933625499130400000) * x + 0.98369508190545300000);
  }
  if (x > -8) {
    if (x > -4)
      return (float)((((0.00217245711583303000 * x + 0.03484829428350620000) * x + 0.22118199801337800000) * x + 0.67049462206469500000) * x + 0.83556950223398500000);
    return (float)((((0.00012398771025456900 * x + 0.00349155785951272000) * x + 0.03727721426017900000) * x + 0.17974997741536900000) * x + 0.33249299994217400000);
  }
  if (x > -16)
    return (float)((((0.00000051741713416603 * x + 0.00002721456879608080) * x + 0.00053418601865636800) * x + 0.00464101989351936000) * x + 0.01507447981459420000);
  return 0;
}

/* Model tables.  Defaults.h:19-39, wrapper.cpp:134-171 (ReadParameters), ProbabilisticModel.h:55-88 (ctor). */
typedef struct {
  float init[3];
  float trans[3][3];
  float match[256][256];
  float ins[256];
} pc_model;

static pc_model g_model;
static int g_model_ready = 0;

static void pc_model_build(void) {
  static const float initDistrib[3] = {0.9588437676f, 0.0205782652f, 0.0205782652f};
  static const float gapOpen[2] = {0.0190259293f, 0.0190259293f};
  static const float gapExtend[2] = {0.3269913495f, 0.3269913495f};
  static const char alphabet[] = "ACGUTN";
  static const float emitSingleDefault[6] = {0.2270790040f, 0.2422080040f, 0.2839320004f, 0.2464679927f, 0.2464679927f, 0.0003124650f};
  static const float emitPairsDefault[6][6] = {
      {0.1487240046f, 0.0184142999f, 0.0361397006f, 0.0238473993f, 0.0238473993f, 0.0000375308f},
      {0.0184142999f, 0.1583919972f, 0.0275536999f, 0.0389291011f, 0.0389291011f, 0.0000815823f},
      {0.0361397006f, 0.0275536999f, 0.1979320049f, 0.0244289003f, 0.0244289003f, 0.0000824765f},
      {0.0238473993f, 0.0389291011f, 0.0244289003f, 0.1557479948f, 0.1557479948f, 0.0000743985f},
      {0.0238473993f, 0.0389291011f, 0.0244289003f, 0.1557479948f, 0.1557479948f, 0.0000743985f},
      {0.0000375308f, 0.0000815823f, 0.0000824765f, 0.0000743985f, 0.0000743985f, 0.0000263252f}};
  static float emitPairs[256][256];
  float emitSingle[256];
  for (int i = 0; i < 256; i++) {
    emitSingle[i] = 1e-5f;                                 /* wrapper.cpp:17,137: VF(256, 1e-5) */
    for (int j = 0; j < 256; j++) emitPairs[i][j] = 1e-10f; /* wrapper.cpp:16,136 */
  }
  for (int i = 0; i < 6; i++) {
    int lo = alphabet[i] + 32, up = alphabet[i]; /* tolower / toupper */
    emitSingle[lo] = emitSingle[up] = emitSingleDefault[i];
    for (int j = 0; j <= i; j++) {
      int lo2 = alphabet[j] + 32, up2 = alphabet[j];
      float v = emitPairsDefault[i][j];
      emitPairs[lo][lo2] = emitPairs[lo][up2] = emitPairs[up][lo2] = emitPairs[up][up2] = v;
      emitPairs[lo2][lo] = emitPairs[lo2][up] = emitPairs[up2][lo] = emitPairs[up2][up] = v;
    }
  }
  /* ProbabilisticModel.h:59-72 */
  float transMat[3][3] = {{0}};
  transMat[0][0] = 1;
  transMat[0][1] = gapOpen[0];
  transMat[0][2] = gapOpen[1];
  transMat[0][0] -= (gapOpen[0] + gapOpen[1]);
  transMat[1][1] = gapExtend[0];
  transMat[2][2] = gapExtend[1];
  transMat[1][2] = 0;
  transMat[2][1] = 0;
  transMat[1][0] = 1 - gapExtend[0];
  transMat[2][0] = 1 - gapExtend[1];
  /* :75-87, LOG(float) == logf */
  for (int i = 0; i < 3; i++) {
    g_model.init[i] = logf(initDistrib[i]);
    for (int j = 0; j < 3; j++) g_model.trans[i][j] = logf(transMat[i][j]);
  }
  for (int i = 0; i < 256; i++) {
    g_model.ins[i] = logf(emitSingle[i]);
    for (int j = 0; j < 256; j++) g_model.match[i][j] = logf(emitPairs[i][j]);
  }
  g_model_ready = 1;
}

/* export the 7-class tables the product host code must reproduce (used by tests only) */
void orc_probcons_tables(float* init3, float* trans9, float* match256x256, float* ins256) {
  if (!g_model_ready) pc_model_build();
  memcpy(init3, g_model.init, sizeof g_model.init);
  memcpy(trans9, g_model.trans, sizeof g_model.trans);
  memcpy(match256x256, g_model.match, sizeof g_model.match);
  memcpy(ins256, g_model.ins, sizeof g_model.ins);
}

/* ProbabilisticModel.h:105-179.  s1/s2 are 0-based here; the reference's iter[i] is s[i-1]. */
static float* pc_forward(const pc_model* m, const unsigned char* s1, int L1, const unsigned char* s2, int L2) {
  size_t n = (size_t)3 * (L1 + 1) * (L2 + 1);
  float* F = (float*)malloc(n * sizeof(float));
  for (size_t i = 0; i < n; i++) F[i] = LOG_ZERO;
  const int W = L2 + 1;
  F[0 + 3 * (1 * W + 1)] = m->init[0] + m->match[s1[0]][s2[0]];
  F[1 + 3 * (1 * W + 0)] = m->init[1] + m->ins[s1[0]];
  F[2 + 3 * (0 * W + 1)] = m->init[2] + m->ins[s2[0]];
  for (int i = 0; i <= L1; i++) {
    unsigned char c1 = (i == 0) ? '~' : s1[i - 1];
    for (int j = 0; j <= L2; j++) {
      unsigned char c2 = (j == 0) ? '~' : s2[j - 1];
      int ij = 3 * (i * W + j), i1j = ij - 3 * W, ij1 = ij - 3, i1j1 = ij - 3 * W - 3;
      if (i > 1 || j > 1) {
        if (i > 0 && j > 0) {
          F[0 + ij] = F[0 + i1j1] + m->trans[0][0];
          for (int k = 1; k < 3; k++) LOG_PLUS_EQUALS(&F[0 + ij], F[k + i1j1] + m->trans[k][0]);
          F[0 + ij] += m->match[c1][c2];
        }
        if (i > 0) F[1 + ij] = m->ins[c1] + LOG_ADD(F[0 + i1j] + m->trans[0][1], F[1 + i1j] + m->trans[1][1]);
        if (j > 0) F[2 + ij] = m->ins[c2] + LOG_ADD(F[0 + ij1] + m->trans[0][2], F[2 + ij1] + m->trans[2][2]);
      }
    }
  }
  return F;
}

/* ProbabilisticModel.h:197-259 */
static float* pc_backward(const pc_model* m, const unsigned char* s1, int L1, const unsigned char* s2, int L2) {
  size_t n = (size_t)3 * (L1 + 1) * (L2 + 1);
  float* B = (float*)malloc(n * sizeof(float));
  for (size_t i = 0; i < n; i++) B[i] = LOG_ZERO;
  const int W = L2 + 1;
  for (int k = 0; k < 3; k++) B[3 * ((L1 + 1) * W - 1) + k] = m->init[k];
  for (int i = L1; i >= 0; i--) {
    unsigned char c1 = (i == L1) ? '~' : s1[i];
    for (int j = L2; j >= 0; j--) {
      unsigned char c2 = (j == L2) ? '~' : s2[j];
      int ij = 3 * (i * W + j), i1j = ij + 3 * W, ij1 = ij + 3, i1j1 = ij + 3 * W + 3;
      if (i < L1 && j < L2) {
        const float ProbXY = B[0 + i1j1] + m->match[c1][c2];
        for (int k = 0; k < 3; k++) LOG_PLUS_EQUALS(&B[k + ij], ProbXY + m->trans[k][0]);
      }
      if (i < L1) {
        LOG_PLUS_EQUALS(&B[0 + ij], B[1 + i1j] + m->ins[c1] + m->trans[0][1]);
        LOG_PLUS_EQUALS(&B[1 + ij], B[1 + i1j] + m->ins[c1] + m->trans[1][1]);
      }
      if (j < L2) {
        LOG_PLUS_EQUALS(&B[0 + ij], B[2 + ij1] + m->ins[c2] + m->trans[0][2]);
        LOG_PLUS_EQUALS(&B[2 + ij], B[2 + ij1] + m->ins[c2] + m->trans[2][2]);
      }
    }
  }
  return B;
}

/* ProbabilisticModel.h:337-403 + wrapper.cpp:120-129 */
int orc_probcons_posterior(const char* s1c, uint32_t L1u, const char* s2c, uint32_t L2u, float th, float* out) {
  if (!g_model_ready) pc_model_build();
  if (L1u == 0 || L2u == 0) return -1;
  const unsigned char* s1 = (const unsigned char*)s1c;
  const unsigned char* s2 = (const unsigned char*)s2c;
  int L1 = (int)L1u, L2 = (int)L2u, W = L2 + 1;
  float* F = pc_forward(&g_model, s1, L1, s2, L2);
  float* B = pc_backward(&g_model, s1, L1, s2, L2);
  float totalF = LOG_ZERO;
  int last = 3 * ((L1 + 1) * W - 1);
  for (int k = 0; k < 3; k++) LOG_PLUS_EQUALS(&totalF, F[k + last] + B[k + last]);
  float totalB = F[0 + 3 * (1 * W + 1)] + B[0 + 3 * (1 * W + 1)];
  LOG_PLUS_EQUALS(&totalB, F[1 + 3 * (1 * W + 0)] + B[1 + 3 * (1 * W + 0)]);
  LOG_PLUS_EQUALS(&totalB, F[2 + 3 * (0 * W + 1)] + B[2 + 3 * (0 * W + 1)]);
  float total = (totalF + totalB) / 2;
  int n = (L1 + 1) * W;
  for (int c = 0; c < n; c++) {
    float v = F[3 * c] + B[3 * c] - total;
    float p = EXP(v < 0.0f ? v : 0.0f); /* min(LOG_ONE, v) */
    out[c] = p;
  }
  out[0] = 0;
  for (int c = 0; c < n; c++) out[c] = (out[c] >= th ? out[c] : 0.0f);
  free(F);
  free(B);
  return n;
}
