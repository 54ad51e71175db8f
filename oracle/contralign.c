/* oracle/contralign.c -- TEST INFRASTRUCTURE (see oracle.h).
 *
 * CPU restatement of the CONTRAlign 5-state pair-CRF match posterior as DAFS uses it
 * (reference src/contralign built with -DRNA=1: states MATCH, INS_X, INS_Y, INS2_X, INS2_Y;
 * 24 RNA weights, src/contralign/Defaults.ipp:389-419).  Loop structure follows the reference
 * (the backward pass is the scatter form) so float rounding is identical.
 * PINNED: bit-exact against oracle/_ref on tests/golden/contralign_mp.npz.
 */
#include "oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define NEG_INF (-2e20f)
enum { MATCH = 0, INS_X = 1, INS_Y = 2, INS2_X = 3, INS2_Y = 4, K = 5 };

/* LogSpace.hpp (contralign copy, identical to contrafold's): :28-60, :74-107, :239-244 */
static inline float Fast_Exp(float x) {
  if (x < (float)(-2.4915033807)) {
    if (x < (float)(-5.8622823336)) {
      if (x < (float)(-9.91152)) return (float)(0);
      return (((float)(0.0000803850) * x + (float)(0.0021627428)) * x + (float)(0.0194708555)) * x + (float)(0.0588080014);
    }
    if (x < (float)(-3.8396630909))
      return (((float)(0.0013889414) * x + (float)(0.0244676474)) * x + (float)(0.1471290604)) * x + (float)(0.3042757740);
    return (((float)(0.0072335607) * x + (float)(0.0906002677)) * x + (float)(0.3983111356)) * x + (float)(0.6245959221);
  }
  if (x < (float)(-0.6725053211)) {
    if (x < (float)(-1.4805375919))
      return (((float)(0.0232410351) * x + (float)(0.2085645908)) * x + (float)(0.6906367911)) * x + (float)(0.8682322329);
    return (((float)(0.0573782771) * x + (float)(0.3580258429)) * x + (float)(0.9121133217)) * x + (float)(0.9793091728);
  }
  if (x < (float)(0))
    return (((float)(0.1199175927) * x + (float)(0.4815668234)) * x + (float)(0.9975991939)) * x + (float)(0.9999505077);
  return (x > (float)(46.052) ? (float)(1e20) : expf(x));
}
static inline float Fast_LogExpPlusOne(float x) {
  if (x < (float)(3.3792499610)) {
    if (x < (float)(1.6320158198)) {
      if (x < (float)(0.6615367791))
        return (((float)(-0.0065591595) * x + (float)(0.1276442762)) * x + (float)(0.4996554598)) * x + (float)(0.6931542306);
      return (((float)(-0.0155157557) * x + (float)(0.1446775699)) * x + (float)(0.4882939746)) * x + (float)(0.6958092989);
    }
    if (x < (float)(2.4912588184))
      return (((float)(-0.0128909247) * x + (float)(0.1301028251)) * x + (float)(0.5150398748)) * x + (float)(0.6795585882);
    return (((float)(-0.0072142647) * x + (float)(0.0877540853)) * x + (float)(0.6208708362)) * x + (float)(0.5909675829);
  }
  if (x < (float)(5.7890710412)) {
    if (x < (float)(4.4261691294))
      return (((float)(-0.0031455354) * x + (float)(0.0467229449)) * x + (float)(0.7592532310)) * x + (float)(0.4348794399);
    return (((float)(-0.0010110698) * x + (float)(0.0185943421)) * x + (float)(0.8831730747)) * x + (float)(0.2523695427);
  }
  if (x < (float)(7.8162726752))
    return (((float)(-0.0001962780) * x + (float)(0.0046084408)) * x + (float)(0.9634431978)) * x + (float)(0.0983148903);
  return (((float)(-0.0000113994) * x + (float)(0.0003734731)) * x + (float)(0.9959107193)) * x + (float)(0.0149855051);
}
static inline void LPE(float* x, float y) {
  float a = *x, b = y;
  if (a < b) { float t = a; a = b; b = t; }
  if (b > (float)(NEG_INF / 2) && a - b < (float)(11.8624794162)) a = Fast_LogExpPlusOne(a - b) + b;
  *x = a;
}

/* 24 RNA weights, Defaults.ipp:393-416, expanded per RegisterParameters (InferenceEngine.ipp:139-226) */
static float score_match[5][5], score_insert[5], score_single[K], score_pair[K][K];
static int ca_ready = 0;
static void ca_build(void) {
  static const float m[10] = {(float)(0.5256508867), (float)(-0.4090640200), (float)(-0.2502759109), (float)(-0.3252306723), (float)(0.6665219366),
                              (float)(-0.3289391181), (float)(-0.1326088918), (float)(0.6684676551), (float)(-0.3565888168), (float)(0.4590520450)};
  /* match_XY with name = lexicographic min: AA AC AG AU CC CG CU GG GU UU */
  int t = 0;
  memset(score_match, 0, sizeof score_match);
  for (int i = 0; i < 4; i++)
    for (int j = i; j < 4; j++) { score_match[i][j] = score_match[j][i] = m[t++]; }
  static const float ins[4] = {(float)(-0.0025219272), (float)(-0.0831389156), (float)(-0.0744397065), (float)(-0.0129005460)};
  for (int i = 0; i < 4; i++) score_insert[i] = ins[i];
  score_insert[4] = 0;
  const float s_match = (float)(0.3959924457), s_insert = (float)(-0.4431756229), s_insert2 = (float)(-0.3488104904);
  score_single[MATCH] = s_match; score_single[INS_X] = score_single[INS_Y] = s_insert; score_single[INS2_X] = score_single[INS2_Y] = s_insert2;
  const float m2m = (float)(2.5057567100), m2i = (float)(-1.2423961130), iext = (float)(1.8676346730), ichg = (float)(-6.9696754440);
  const float m2i2 = (float)(0.1970448791), i2ext = (float)(1.0140265830), i2chg = (float)(-7.3469687820);
  memset(score_pair, 0, sizeof score_pair);
  score_pair[MATCH][MATCH] = m2m;
  score_pair[MATCH][INS_X] = score_pair[MATCH][INS_Y] = score_pair[INS_X][MATCH] = score_pair[INS_Y][MATCH] = m2i;
  score_pair[INS_X][INS_X] = score_pair[INS_Y][INS_Y] = iext;
  score_pair[INS_X][INS_Y] = score_pair[INS_Y][INS_X] = ichg;
  score_pair[MATCH][INS2_X] = score_pair[MATCH][INS2_Y] = score_pair[INS2_X][MATCH] = score_pair[INS2_Y][MATCH] = m2i2;
  score_pair[INS2_X][INS2_X] = score_pair[INS2_Y][INS2_Y] = i2ext;
  score_pair[INS2_X][INS2_Y] = score_pair[INS2_Y][INS2_X] = i2chg;
  ca_ready = 1;
}
void orc_contralign_tables(float* match25, float* insert5, float* single5, float* pair25) {
  if (!ca_ready) ca_build();
  memcpy(match25, score_match, sizeof score_match);
  memcpy(insert5, score_insert, sizeof score_insert);
  memcpy(single5, score_single, sizeof score_single);
  memcpy(pair25, score_pair, sizeof score_pair);
}

static int cmap(unsigned char c) { /* InferenceEngine ctor: case-insensitive "ACGU", else 4 */
  switch (c) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'U': case 'u': return 3;
    default: return 4;
  }
}

typedef struct { int LX, LY; const int *x, *y; } eng;
/* Score* with no alignment constraints (aligned_to = UNKNOWN), InferenceEngine.ipp:446-782 */
static inline float ScoreMatch(const eng* e, int i, int j, int s) {
  return (float)(0) + score_match[e->x[i]][e->y[j]] + score_single[MATCH] + (i != 1 || j != 1 ? score_pair[s][MATCH] : (float)(0));
}
static inline float ScoreInsertX(const eng* e, int i, int j, int s) {
  return (float)(0) + score_insert[e->x[i]] + score_single[INS_X] + (i != 1 || j != 0 ? score_pair[s][INS_X] : (float)(0));
}
static inline float ScoreInsert2X(const eng* e, int i, int j, int s) {
  return (float)(0) + score_insert[e->x[i]] + score_single[INS2_X] + (i != 1 || j != 0 ? score_pair[s][INS2_X] : (float)(0));
}
static inline float ScoreInsertY(const eng* e, int i, int j, int s) {
  return (float)(0) + score_insert[e->y[j]] + score_single[INS_Y] + (i != 0 || j != 1 ? score_pair[s][INS_Y] : (float)(0));
}
static inline float ScoreInsert2Y(const eng* e, int i, int j, int s) {
  return (float)(0) + score_insert[e->y[j]] + score_single[INS2_Y] + (i != 0 || j != 1 ? score_pair[s][INS2_Y] : (float)(0));
}

int orc_contralign_posterior(const char* s1, uint32_t L1, const char* s2, uint32_t L2, float th, float* out) {
  if (!ca_ready) ca_build();
  if (L1 == 0 || L2 == 0) return -1;
  const int LX = (int)L1, LY = (int)L2, W = LY + 1, SIZE = (LX + 1) * (LY + 1);
  int* x = (int*)malloc((LX + 1) * sizeof(int));
  int* y = (int*)malloc((LY + 1) * sizeof(int));
  x[0] = 4; y[0] = 4;
  for (int i = 1; i <= LX; i++) x[i] = cmap((unsigned char)s1[i - 1]);
  for (int j = 1; j <= LY; j++) y[j] = cmap((unsigned char)s2[j - 1]);
  eng E = {LX, LY, x, y};
  const eng* e = &E;
  float* Ff[K];
  float* Fb[K];
  for (int k = 0; k < K; k++) {
    Ff[k] = (float*)malloc((size_t)SIZE * sizeof(float));
    Fb[k] = (float*)malloc((size_t)SIZE * sizeof(float));
    for (int c = 0; c < SIZE; c++) { Ff[k][c] = NEG_INF; Fb[k][c] = NEG_INF; }
    Ff[k][0] = (float)(0);
    Fb[k][SIZE - 1] = (float)(0);
  }
  /* ComputeForward, InferenceEngine.ipp:999-1070 */
  for (int i = 1; i <= LX; i++) LPE(&Ff[INS_X][i * W + 0], Ff[INS_X][(i - 1) * W + 0] + ScoreInsertX(e, i, 0, INS_X));
  for (int j = 1; j <= LY; j++) LPE(&Ff[INS_Y][0 * W + j], Ff[INS_Y][0 * W + (j - 1)] + ScoreInsertY(e, 0, j, INS_Y));
  for (int i = 1; i <= LX; i++) LPE(&Ff[INS2_X][i * W + 0], Ff[INS2_X][(i - 1) * W + 0] + ScoreInsert2X(e, i, 0, INS2_X));
  for (int j = 1; j <= LY; j++) LPE(&Ff[INS2_Y][0 * W + j], Ff[INS2_Y][0 * W + (j - 1)] + ScoreInsert2Y(e, 0, j, INS2_Y));
  for (int i = 1; i <= LX; i++)
    for (int j = 1; j <= LY; j++) {
      const int ij = i * W + j, i1j = ij - W, ij1 = ij - 1, i1j1 = ij - W - 1;
      LPE(&Ff[MATCH][ij], Ff[MATCH][i1j1] + ScoreMatch(e, i, j, MATCH));
      if (i > 1 || j > 1) {
        LPE(&Ff[MATCH][ij], Ff[INS_X][i1j1] + ScoreMatch(e, i, j, INS_X));
        LPE(&Ff[MATCH][ij], Ff[INS_Y][i1j1] + ScoreMatch(e, i, j, INS_Y));
        LPE(&Ff[MATCH][ij], Ff[INS2_X][i1j1] + ScoreMatch(e, i, j, INS2_X));
        LPE(&Ff[MATCH][ij], Ff[INS2_Y][i1j1] + ScoreMatch(e, i, j, INS2_Y));
      }
      LPE(&Ff[INS_X][ij], Ff[MATCH][i1j] + ScoreInsertX(e, i, j, MATCH));
      LPE(&Ff[INS_X][ij], Ff[INS_X][i1j] + ScoreInsertX(e, i, j, INS_X));
      LPE(&Ff[INS_X][ij], Ff[INS_Y][i1j] + ScoreInsertX(e, i, j, INS_Y));
      LPE(&Ff[INS_Y][ij], Ff[MATCH][ij1] + ScoreInsertY(e, i, j, MATCH));
      LPE(&Ff[INS_Y][ij], Ff[INS_X][ij1] + ScoreInsertY(e, i, j, INS_X));
      LPE(&Ff[INS_Y][ij], Ff[INS_Y][ij1] + ScoreInsertY(e, i, j, INS_Y));
      LPE(&Ff[INS2_X][ij], Ff[MATCH][i1j] + ScoreInsert2X(e, i, j, MATCH));
      LPE(&Ff[INS2_X][ij], Ff[INS2_X][i1j] + ScoreInsert2X(e, i, j, INS2_X));
      LPE(&Ff[INS2_X][ij], Ff[INS2_Y][i1j] + ScoreInsert2X(e, i, j, INS2_Y));
      LPE(&Ff[INS2_Y][ij], Ff[MATCH][ij1] + ScoreInsert2Y(e, i, j, MATCH));
      LPE(&Ff[INS2_Y][ij], Ff[INS2_X][ij1] + ScoreInsert2Y(e, i, j, INS2_X));
      LPE(&Ff[INS2_Y][ij], Ff[INS2_Y][ij1] + ScoreInsert2Y(e, i, j, INS2_Y));
    }
  /* ComputeBackward, :1079-1150 */
  for (int i = LX; i >= 1; i--)
    for (int j = LY; j >= 1; j--) {
      const int ij = i * W + j, i1j = ij - W, ij1 = ij - 1, i1j1 = ij - W - 1;
      LPE(&Fb[MATCH][i1j1], Fb[MATCH][ij] + ScoreMatch(e, i, j, MATCH));
      if (i > 1 || j > 1) {
        LPE(&Fb[INS_X][i1j1], Fb[MATCH][ij] + ScoreMatch(e, i, j, INS_X));
        LPE(&Fb[INS_Y][i1j1], Fb[MATCH][ij] + ScoreMatch(e, i, j, INS_Y));
        LPE(&Fb[INS2_X][i1j1], Fb[MATCH][ij] + ScoreMatch(e, i, j, INS2_X));
        LPE(&Fb[INS2_Y][i1j1], Fb[MATCH][ij] + ScoreMatch(e, i, j, INS2_Y));
      }
      LPE(&Fb[MATCH][i1j], Fb[INS_X][ij] + ScoreInsertX(e, i, j, MATCH));
      LPE(&Fb[INS_X][i1j], Fb[INS_X][ij] + ScoreInsertX(e, i, j, INS_X));
      LPE(&Fb[INS_Y][i1j], Fb[INS_X][ij] + ScoreInsertX(e, i, j, INS_Y));
      LPE(&Fb[MATCH][ij1], Fb[INS_Y][ij] + ScoreInsertY(e, i, j, MATCH));
      LPE(&Fb[INS_X][ij1], Fb[INS_Y][ij] + ScoreInsertY(e, i, j, INS_X));
      LPE(&Fb[INS_Y][ij1], Fb[INS_Y][ij] + ScoreInsertY(e, i, j, INS_Y));
      LPE(&Fb[MATCH][i1j], Fb[INS2_X][ij] + ScoreInsert2X(e, i, j, MATCH));
      LPE(&Fb[INS2_X][i1j], Fb[INS2_X][ij] + ScoreInsert2X(e, i, j, INS2_X));
      LPE(&Fb[INS2_Y][i1j], Fb[INS2_X][ij] + ScoreInsert2X(e, i, j, INS2_Y));
      LPE(&Fb[MATCH][ij1], Fb[INS2_Y][ij] + ScoreInsert2Y(e, i, j, MATCH));
      LPE(&Fb[INS2_X][ij1], Fb[INS2_Y][ij] + ScoreInsert2Y(e, i, j, INS2_X));
      LPE(&Fb[INS2_Y][ij1], Fb[INS2_Y][ij] + ScoreInsert2Y(e, i, j, INS2_Y));
    }
  /* (the four border loops of ComputeBackward only touch row 0 / column 0, which nothing below reads) */
  /* ComputeForwardLogPartitionCoefficient, :1164-1170 */
  float Z = Ff[MATCH][SIZE - 1];
  for (int k = 1; k < K; k++) LPE(&Z, Ff[k][SIZE - 1]);
  /* ComputePosterior, :1279-1317 + GetPosterior(th) :1428-1438 */
  for (int c = 0; c < SIZE; c++) out[c] = (float)(0);
  for (int i = 1; i <= LX; i++)
    for (int j = 1; j <= LY; j++) {
      const int ij = i * W + j, i1j1 = ij - W - 1;
      float p = (float)(0);
      for (int k = 0; k < K; k++)
        if (k == MATCH || i > 1 || j > 1) p += Fast_Exp(Ff[k][i1j1] + ScoreMatch(e, i, j, k) + Fb[MATCH][ij] - Z);
      float m = p < (float)(0) ? (float)(0) : p;
      out[ij] = ((float)(1) < m) ? (float)(1) : m;
    }
  for (int c = 0; c < SIZE; c++) out[c] = (out[c] >= th ? out[c] : (float)(0));
  for (int k = 0; k < K; k++) { free(Ff[k]); free(Fb[k]); }
  free(x); free(y);
  return SIZE;
}
