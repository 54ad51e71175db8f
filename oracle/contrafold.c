/* oracle/contrafold.c -- TEST INFRASTRUCTURE (see oracle.h).
 *
 * CPU restatement of the CONTRAfold base-pairing posterior as DAFS uses it
 * (reference src/contrafold, CONTRAfold<float>(canonical_only=true, max_bp_dist=0), live feature
 * switches of src/contrafold/Config.hpp:156-179: no helix-length / isolated-base-pair states, so
 * the grammar is FC / FM / FM1 / F5).  It follows the reference's loop structure (the outside
 * pass is the scatter form) so that float rounding is the same operation for operation.
 * PINNED: bit-exact against oracle/_ref on tests/golden/contrafold_post.npz (10 tRNAs, synthetic
 * L=80/150, short and odd sequences, one constrained fold).
 */
#include "oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "contrafold_params.h"

#define NEG_INF (-2e20f)          /* LogSpace.hpp:13, narrowed to RealT=float */
#define C_MAX_SINGLE_LENGTH 30    /* Config.hpp:213 */
#define D_MAX_HAIRPIN_LENGTH 30
#define UNPAIRED 0                /* SStruct.cpp:14-15 */
#define UNKNOWN (-1)

/* LogSpace.hpp:28-60 */
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

/* LogSpace.hpp:74-107 */
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

/* LogSpace.hpp:239-244 */
static inline void LPE(float* x, float y) {
  float a = *x, b = y;
  if (a < b) { float t = a; a = b; b = t; }
  if (b > (float)(NEG_INF / 2) && a - b < (float)(11.8624794162)) a = Fast_LogExpPlusOne(a - b) + b;
  *x = a;
}

typedef struct {
  int L, SIZE;
  int* s;
  int* offset;
  int* allow_unpaired_position;
  int* allow_unpaired;
  int* allow_paired;
  float cache_hairpin[D_MAX_HAIRPIN_LENGTH + 1];
  float cache_single[C_MAX_SINGLE_LENGTH + 1][C_MAX_SINGLE_LENGTH + 1];
  float *F5i, *FCi, *FMi, *FM1i, *F5o, *FCo, *FMo, *FM1o, *posterior;
} cf_t;

static int char_map(unsigned char c) { /* InferenceEngine.ipp ctor: case-insensitive "ACGU", else M=4 */
  switch (c) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'U': case 'u': return 3;
    default: return 4;
  }
}
static int is_complementary(int a, int b) { /* AU, GU, CG and their reverses */
  return (a == 0 && b == 3) || (a == 3 && b == 0) || (a == 2 && b == 3) || (a == 3 && b == 2) || (a == 1 && b == 2) || (a == 2 && b == 1);
}

#define OFF(i) (e->offset[i])
#define S(i) (e->s[i])

static inline float ScoreJunctionA(const cf_t* e, int i, int j) {
  return (float)(0) + cf_helix_closing[S(i)][S(j + 1)] + (i < e->L ? cf_dangle_left[S(i)][S(j + 1)][S(i + 1)] : (float)(0)) +
         (j > 0 ? cf_dangle_right[S(i)][S(j + 1)][S(j)] : (float)(0));
}
static inline float ScoreJunctionB(const cf_t* e, int i, int j) {
  return (float)(0) + cf_helix_closing[S(i)][S(j + 1)] + cf_terminal_mismatch[S(i)][S(j + 1)][S(i + 1)][S(j)];
}
static inline float ScoreBasePair(const cf_t* e, int i, int j) { return (float)(0) + cf_base_pair[S(i)][S(j)]; }
static inline float ScoreHelixStacking(const cf_t* e, int i, int j) { return cf_helix_stacking[S(i)][S(j)][S(i + 1)][S(j - 1)]; }
static inline float ScoreHairpin(const cf_t* e, int i, int j) {
  int d = j - i < D_MAX_HAIRPIN_LENGTH ? j - i : D_MAX_HAIRPIN_LENGTH;
  return ((float)(0)) + ScoreJunctionB(e, i, j) + e->cache_hairpin[d];
}
static inline float ScoreSingleNucleotides(const cf_t* e, int i, int j, int p, int q) {
  const int l1 = p - i, l2 = j - q;
  return ((float)(0)) + ((float)(0)) + (l1 == 0 && l2 == 1 ? cf_bulge_0x1_nucleotides[S(j)] : (float)(0)) +
         (l1 == 1 && l2 == 0 ? cf_bulge_1x0_nucleotides[S(i + 1)] : (float)(0)) +
         (l1 == 1 && l2 == 1 ? cf_internal_1x1_nucleotides[S(i + 1)][S(j)] : (float)(0));
}
#define ScoreMultiBase() (cf_multi_base)
#define ScoreMultiPaired() (cf_multi_paired)
#define ScoreMultiUnpaired(i) (cf_multi_unpaired + ((float)(0)))
#define ScoreExternalPaired() (cf_external_paired)
#define ScoreExternalUnpaired(i) (cf_external_unpaired + ((float)(0)))

/* InitializeCache, InferenceEngine.ipp:1106-1335 (live parts) */
static void initialize_cache(cf_t* e) {
  e->cache_hairpin[0] = cf_hairpin_length_at_least[0];
  for (int i = 1; i <= D_MAX_HAIRPIN_LENGTH; i++) e->cache_hairpin[i] = e->cache_hairpin[i - 1] + cf_hairpin_length_at_least[i];
  float bulge[31], internal[31], sym[16], asym[29];
  bulge[0] = cf_bulge_length_at_least[0];
  for (int i = 1; i <= 30; i++) bulge[i] = bulge[i - 1] + cf_bulge_length_at_least[i];
  internal[0] = cf_internal_length_at_least[0];
  for (int i = 1; i <= 30; i++) internal[i] = internal[i - 1] + cf_internal_length_at_least[i];
  sym[0] = cf_internal_symmetric_length_at_least[0];
  for (int i = 1; i <= 15; i++) sym[i] = sym[i - 1] + cf_internal_symmetric_length_at_least[i];
  asym[0] = cf_internal_asymmetry_at_least[0];
  for (int i = 1; i <= 28; i++) asym[i] = asym[i - 1] + cf_internal_asymmetry_at_least[i];
  for (int l1 = 0; l1 <= C_MAX_SINGLE_LENGTH; l1++)
    for (int l2 = 0; l1 + l2 <= C_MAX_SINGLE_LENGTH; l2++) {
      float v = (float)(0);
      if (l1 == 0 && l2 == 0) { e->cache_single[l1][l2] = v; continue; }
      if (l1 == 0 || l2 == 0) {
        v += bulge[l1 + l2 < 30 ? l1 + l2 : 30];
      } else {
        if (l1 <= 4 && l2 <= 4) v += cf_internal_explicit[l1][l2];
        v += internal[l1 + l2 < 30 ? l1 + l2 : 30];
        if (l1 == l2) v += sym[l1 < 15 ? l1 : 15];
        int d = l1 > l2 ? l1 - l2 : l2 - l1;
        v += asym[d < 28 ? d : 28];
      }
      e->cache_single[l1][l2] = v;
    }
}

/* LoadSequence (:947-1097) + UseConstraints (:1870-1902) */
static int load_sequence(cf_t* e, const char* seq, int L, const char* constraint) {
  e->L = L;
  e->SIZE = (L + 1) * (L + 2) / 2;
  e->s = (int*)malloc((L + 2) * sizeof(int));
  e->offset = (int*)malloc((L + 2) * sizeof(int));
  e->allow_unpaired_position = (int*)malloc((L + 2) * sizeof(int));
  e->allow_unpaired = (int*)malloc((size_t)e->SIZE * sizeof(int));
  e->allow_paired = (int*)malloc((size_t)e->SIZE * sizeof(int));
  e->s[0] = 4;
  for (int i = 1; i <= L; i++) e->s[i] = char_map((unsigned char)seq[i - 1]);
  e->s[L + 1] = 4; /* never read by the reference (all uses are guarded); keeps our reads in bounds */
  const int N = L + 1;
  for (int i = 0; i <= L; i++) {
    e->offset[i] = i * (N + N - i - 1) / 2;
    e->allow_unpaired_position[i] = 1;
  }
  for (int i = 0; i < e->SIZE; i++) { e->allow_unpaired[i] = 1; e->allow_paired[i] = 1; }
  for (int i = 0; i <= L; i++) { e->allow_paired[OFF(0) + i] = 0; e->allow_paired[OFF(i) + i] = 0; }
  for (int i = 1; i <= L; i++)
    for (int j = i + 1; j <= L; j++)
      if (!is_complementary(S(i), S(j))) e->allow_paired[OFF(i) + j] = 0;
  if (constraint) {
    /* SStruct::ConvertParensToMapping, SStruct.cpp:389-417 ('-' is read as '.', :365-381) */
    int* mapping = (int*)malloc((L + 1) * sizeof(int));
    int* stack = (int*)malloc((L + 1) * sizeof(int));
    int sp = 0;
    for (int i = 0; i <= L; i++) mapping[i] = UNKNOWN;
    for (int i = 1; i <= L; i++) {
      char c = constraint[i - 1];
      if (c == '?') continue;
      if (c == '.' || c == '-') mapping[i] = UNPAIRED;
      else if (c == '(') stack[sp++] = i;
      else if (c == ')') {
        if (!sp) { free(mapping); free(stack); return -1; }
        mapping[i] = stack[sp - 1];
        mapping[stack[sp - 1]] = i;
        --sp;
      } else { free(mapping); free(stack); return -1; }
    }
    if (sp) { free(mapping); free(stack); return -1; }
    for (int i = 1; i <= L; i++) e->allow_unpaired_position[i] = (mapping[i] == UNKNOWN || mapping[i] == UNPAIRED);
    for (int i = 0; i <= L; i++) {
      e->allow_unpaired[OFF(i) + i] = 1;
      e->allow_paired[OFF(i) + i] = 0;
      for (int j = i + 1; j <= L; j++) {
        e->allow_unpaired[OFF(i) + j] = e->allow_unpaired[OFF(i) + j - 1] && e->allow_unpaired_position[j];
        e->allow_paired[OFF(i) + j] = (i > 0 && (mapping[i] == UNKNOWN || mapping[i] == j) && (mapping[j] == UNKNOWN || mapping[j] == i) &&
                                       is_complementary(S(i), S(j)));
      }
    }
    free(mapping);
    free(stack);
  }
  return 0;
}

static float* falloc(int n, float v) {
  float* p = (float*)malloc((size_t)(n > 0 ? n : 1) * sizeof(float));
  for (int i = 0; i < n; i++) p[i] = v;
  return p;
}

/* ComputeInside, InferenceEngine.ipp:3356-3722 (max_bp_dist == 0 branch) */
static void compute_inside(cf_t* e) {
  const int L = e->L;
  e->F5i = falloc(L + 1, NEG_INF);
  e->FCi = falloc(e->SIZE, NEG_INF);
  e->FMi = falloc(e->SIZE, NEG_INF);
  e->FM1i = falloc(e->SIZE, NEG_INF);
  float *FCi = e->FCi, *FMi = e->FMi, *FM1i = e->FM1i, *F5i = e->F5i;
  for (int i = L; i >= 0; i--) {
    for (int j = i; j <= L; j++) {
      float FM2i = NEG_INF;
      if (i + 2 <= j)
        for (int k = i + 1; k < j; k++) LPE(&FM2i, FM1i[OFF(i) + k] + FMi[OFF(k) + j]);
      if (0 < i && j < L && e->allow_paired[OFF(i) + j + 1]) {
        float sum_i = NEG_INF;
        if (e->allow_unpaired[OFF(i) + j] && j - i >= 0) LPE(&sum_i, ScoreHairpin(e, i, j));
        {
          float score_helix = (i + 2 <= j ? ScoreBasePair(e, i + 1, j) + ScoreHelixStacking(e, i, j + 1) : 0);
          float score_other = ScoreJunctionB(e, i, j);
          const int pmax = i + C_MAX_SINGLE_LENGTH < j ? i + C_MAX_SINGLE_LENGTH : j;
          for (int p = i; p <= pmax; p++) {
            if (p > i && !e->allow_unpaired_position[p]) break;
            int q_min = p + 2 > p - i + j - C_MAX_SINGLE_LENGTH ? p + 2 : p - i + j - C_MAX_SINGLE_LENGTH;
            const float* FCptr = &FCi[OFF(p + 1) - 1];
            for (int q = j; q >= q_min; q--) {
              if (q < j && !e->allow_unpaired_position[q + 1]) break;
              if (!e->allow_paired[OFF(p + 1) + q]) continue;
              float score = (p == i && q == j) ? (score_helix + FCptr[q])
                                               : (score_other + e->cache_single[p - i][j - q] + FCptr[q] + ScoreBasePair(e, p + 1, q) +
                                                  ScoreJunctionB(e, q, p) + ScoreSingleNucleotides(e, i, j, p, q));
              LPE(&sum_i, score);
            }
          }
        }
        LPE(&sum_i, FM2i + ScoreJunctionA(e, i, j) + ScoreMultiPaired() + ScoreMultiBase());
        FCi[OFF(i) + j] = sum_i;
      }
      if (0 < i && i + 2 <= j && j < L) {
        float sum_i = NEG_INF;
        if (e->allow_paired[OFF(i + 1) + j])
          LPE(&sum_i, FCi[OFF(i + 1) + j - 1] + ScoreJunctionA(e, j, i) + ScoreMultiPaired() + ScoreBasePair(e, i + 1, j));
        if (e->allow_unpaired_position[i + 1]) LPE(&sum_i, FM1i[OFF(i + 1) + j] + ScoreMultiUnpaired(i + 1));
        FM1i[OFF(i) + j] = sum_i;
      }
      if (0 < i && i + 2 <= j && j < L) {
        float sum_i = NEG_INF;
        LPE(&sum_i, FM2i);
        if (e->allow_unpaired_position[j]) LPE(&sum_i, FMi[OFF(i) + j - 1] + ScoreMultiUnpaired(j));
        LPE(&sum_i, FM1i[OFF(i) + j]);
        FMi[OFF(i) + j] = sum_i;
      }
    }
  }
  F5i[0] = (float)(0);
  for (int j = 1; j <= L; j++) {
    float sum_i = NEG_INF;
    if (e->allow_unpaired_position[j]) LPE(&sum_i, F5i[j - 1] + ScoreExternalUnpaired(j));
    for (int k = 0; k < j; k++)
      if (e->allow_paired[OFF(k + 1) + j])
        LPE(&sum_i, F5i[k] + FCi[OFF(k + 1) + j - 1] + ScoreExternalPaired() + ScoreBasePair(e, k + 1, j) + ScoreJunctionA(e, j, k));
    F5i[j] = sum_i;
  }
}

/* ComputeOutside, InferenceEngine.ipp:3731-4080 */
static void compute_outside(cf_t* e) {
  const int L = e->L;
  e->F5o = falloc(L + 1, NEG_INF);
  e->FCo = falloc(e->SIZE, NEG_INF);
  e->FMo = falloc(e->SIZE, NEG_INF);
  e->FM1o = falloc(e->SIZE, NEG_INF);
  float *FCi = e->FCi, *FMi = e->FMi, *FM1i = e->FM1i, *F5i = e->F5i;
  float *FCo = e->FCo, *FMo = e->FMo, *FM1o = e->FM1o, *F5o = e->F5o;
  F5o[L] = (float)(0);
  for (int j = L; j >= 1; j--) {
    if (e->allow_unpaired_position[j]) LPE(&F5o[j - 1], F5o[j] + ScoreExternalUnpaired(j));
    for (int k = 0; k < j; k++)
      if (e->allow_paired[OFF(k + 1) + j]) {
        float temp = F5o[j] + ScoreExternalPaired() + ScoreBasePair(e, k + 1, j) + ScoreJunctionA(e, j, k);
        LPE(&F5o[k], temp + FCi[OFF(k + 1) + j - 1]);
        LPE(&FCo[OFF(k + 1) + j - 1], temp + F5i[k]);
      }
  }
  for (int i = 0; i <= L; i++) {
    for (int j = L; j >= i; j--) {
      float FM2o = NEG_INF;
      if (0 < i && i + 2 <= j && j < L) {
        LPE(&FM2o, FMo[OFF(i) + j]);
        if (e->allow_unpaired_position[j]) LPE(&FMo[OFF(i) + j - 1], FMo[OFF(i) + j] + ScoreMultiUnpaired(j));
        LPE(&FM1o[OFF(i) + j], FMo[OFF(i) + j]);
      }
      if (0 < i && i + 2 <= j && j < L) {
        if (e->allow_paired[OFF(i + 1) + j])
          LPE(&FCo[OFF(i + 1) + j - 1], FM1o[OFF(i) + j] + ScoreJunctionA(e, j, i) + ScoreMultiPaired() + ScoreBasePair(e, i + 1, j));
        if (e->allow_unpaired_position[i + 1]) LPE(&FM1o[OFF(i + 1) + j], FM1o[OFF(i) + j] + ScoreMultiUnpaired(i + 1));
      }
      if (0 < i && j < L && e->allow_paired[OFF(i) + j + 1]) {
        {
          float score_helix = (i + 2 <= j ? FCo[OFF(i) + j] + ScoreBasePair(e, i + 1, j) + ScoreHelixStacking(e, i, j + 1) : 0);
          float score_other = FCo[OFF(i) + j] + ScoreJunctionB(e, i, j);
          const int pmax = i + C_MAX_SINGLE_LENGTH < j ? i + C_MAX_SINGLE_LENGTH : j;
          for (int p = i; p <= pmax; p++) {
            if (p > i && !e->allow_unpaired_position[p]) break;
            int q_min = p + 2 > p - i + j - C_MAX_SINGLE_LENGTH ? p + 2 : p - i + j - C_MAX_SINGLE_LENGTH;
            float* FCptr = &FCo[OFF(p + 1) - 1];
            for (int q = j; q >= q_min; q--) {
              if (q < j && !e->allow_unpaired_position[q + 1]) break;
              if (!e->allow_paired[OFF(p + 1) + q]) continue;
              LPE(&FCptr[q], (p == i && q == j) ? score_helix
                                                : score_other + e->cache_single[p - i][j - q] + ScoreBasePair(e, p + 1, q) +
                                                      ScoreJunctionB(e, q, p) + ScoreSingleNucleotides(e, i, j, p, q));
            }
          }
        }
        LPE(&FM2o, FCo[OFF(i) + j] + ScoreJunctionA(e, i, j) + ScoreMultiPaired() + ScoreMultiBase());
      }
      if (i + 2 <= j)
        for (int k = i + 1; k < j; k++) {
          LPE(&FM1o[OFF(i) + k], FM2o + FMi[OFF(k) + j]);
          LPE(&FMo[OFF(k) + j], FM2o + FM1i[OFF(i) + k]);
        }
    }
  }
}

/* ComputePosterior, InferenceEngine.ipp:4498-4821 */
static void compute_posterior(cf_t* e) {
  const int L = e->L;
  e->posterior = falloc(e->SIZE, (float)(0));
  float* posterior = e->posterior;
  float *FCi = e->FCi, *F5i = e->F5i, *FCo = e->FCo, *FM1o = e->FM1o, *F5o = e->F5o;
  const float Z = F5i[L];
  for (int i = L; i >= 0; i--) {
    for (int j = i; j <= L; j++) {
      if (0 < i && j < L && e->allow_paired[OFF(i) + j + 1]) {
        float outside = FCo[OFF(i) + j] - Z;
        float score_helix = (i + 2 <= j ? outside + ScoreBasePair(e, i + 1, j) + ScoreHelixStacking(e, i, j + 1) : 0);
        float score_other = outside + ScoreJunctionB(e, i, j);
        const int pmax = i + C_MAX_SINGLE_LENGTH < j ? i + C_MAX_SINGLE_LENGTH : j;
        for (int p = i; p <= pmax; p++) {
          if (p > i && !e->allow_unpaired_position[p]) break;
          int q_min = p + 2 > p - i + j - C_MAX_SINGLE_LENGTH ? p + 2 : p - i + j - C_MAX_SINGLE_LENGTH;
          const float* FCptr = &FCi[OFF(p + 1) - 1];
          for (int q = j; q >= q_min; q--) {
            if (q < j && !e->allow_unpaired_position[q + 1]) break;
            if (!e->allow_paired[OFF(p + 1) + q]) continue;
            posterior[OFF(p + 1) + q] +=
                Fast_Exp(p == i && q == j ? score_helix + FCptr[q]
                                          : score_other + e->cache_single[p - i][j - q] + FCptr[q] + ScoreBasePair(e, p + 1, q) +
                                                ScoreJunctionB(e, q, p) + ScoreSingleNucleotides(e, i, j, p, q));
          }
        }
      }
      if (0 < i && i + 2 <= j && j < L) {
        if (e->allow_paired[OFF(i + 1) + j])
          posterior[OFF(i + 1) + j] += Fast_Exp(FM1o[OFF(i) + j] + FCi[OFF(i + 1) + j - 1] + ScoreJunctionA(e, j, i) + ScoreMultiPaired() +
                                                ScoreBasePair(e, i + 1, j) - Z);
      }
    }
  }
  for (int j = 1; j <= L; j++) {
    float outside = F5o[j] - Z;
    for (int k = 0; k < j; k++)
      if (e->allow_paired[OFF(k + 1) + j])
        posterior[OFF(k + 1) + j] += Fast_Exp(outside + F5i[k] + FCi[OFF(k + 1) + j - 1] + ScoreExternalPaired() + ScoreBasePair(e, k + 1, j) +
                                              ScoreJunctionA(e, j, k));
  }
  for (int i = 1; i <= L; i++)
    for (int j = i + 1; j <= L; j++) { /* Clip = min(max(x, 0), 1), Utilities.ipp:136 */
      float x = posterior[OFF(i) + j];
      float m = x < (float)(0) ? (float)(0) : x; /* std::max(x, lower): (x < lower) ? lower : x */
      posterior[OFF(i) + j] = ((float)(1) < m) ? (float)(1) : m; /* std::min(m, upper): (upper < m) ? upper : m */
    }
}

static void cf_free(cf_t* e) {
  free(e->s); free(e->offset); free(e->allow_unpaired_position); free(e->allow_unpaired); free(e->allow_paired);
  free(e->F5i); free(e->FCi); free(e->FMi); free(e->FM1i); free(e->F5o); free(e->FCo); free(e->FMo); free(e->FM1o); free(e->posterior);
}

/* CONTRAfold<float>::Impl::ComputePosterior, wrapper.cpp:181-200 + GetPosterior(0.0) (:5614-5621) */
int orc_contrafold_posterior(const char* seq, uint32_t Lu, const char* constraint, float* out) {
  cf_t e;
  memset(&e, 0, sizeof e);
  if (constraint && !constraint[0]) constraint = NULL; /* empty constraint string = unconstrained (wrapper.cpp:188) */
  if (load_sequence(&e, seq, (int)Lu, constraint)) { cf_free(&e); return -1; }
  initialize_cache(&e);
  compute_inside(&e);
  compute_outside(&e);
  compute_posterior(&e);
  for (int i = 0; i < e.SIZE; i++) out[i] = (e.posterior[i] >= 0.0f ? e.posterior[i] : (float)(0));
  int n = e.SIZE;
  cf_free(&e);
  return n;
}

float orc_contrafold_logz(const char* seq, uint32_t Lu) {
  cf_t e;
  memset(&e, 0, sizeof e);
  if (load_sequence(&e, seq, (int)Lu, NULL)) return 0.0f / 0.0f;
  initialize_cache(&e);
  compute_inside(&e);
  float z = e.F5i[e.L];
  cf_free(&e);
  return z;
}

/* CONTRAfold::calculate, src/fold.cpp:174-207: rows (i-1) -> (j-1, p) for p > th, i != 0 */
int orc_fold_calculate(const char* seq, uint32_t L, const char* constraint, float th, uint32_t* rowptr, uint32_t* col, float* val) {
  float* post = (float*)malloc((size_t)(L + 1) * (L + 2) / 2 * sizeof(float));
  int rc = orc_contrafold_posterior(seq, L, constraint, post);
  if (rc < 0) { free(post); return rc; }
  uint32_t n = 0, k = 0;
  for (uint32_t i = 0; i != L + 1; ++i) {
    if (i != 0) rowptr[i - 1] = n;
    for (uint32_t j = i; j != L + 1; ++j, ++k)
      if (i != 0 && post[k] > th) { col[n] = j - 1; val[n] = post[k]; ++n; }
  }
  rowptr[L] = n;
  free(post);
  return (int)n;
}
