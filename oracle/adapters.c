/* oracle/adapters.c -- TEST INFRASTRUCTURE (see oracle.h).
 * Dense -> sparse adapters of the reference plugin layer. */
#include "oracle.h"
#include <stdlib.h>
#include <string.h>

void orc_csr_free(orc_csr* m) {
  if (!m) return;
  free(m->rowptr); free(m->col); free(m->val);
  m->rowptr = NULL; m->col = NULL; m->val = NULL; m->nrow = 0;
}

/* ProbCons::calculate / CONTRAlign::calculate, src/align.cpp:60-79 and :87-106:
 * keep posterior[(L2+1)*(i+1)+(j+1)] > th, rows ascending j. */
int orc_align_calculate(int model, const char* s1, uint32_t L1, const char* s2, uint32_t L2, float th,
                        uint32_t* rowptr, uint32_t* col, float* val) {
  float* post = (float*)malloc((size_t)(L1 + 1) * (L2 + 1) * sizeof(float));
  int rc = model == 0 ? orc_probcons_posterior(s1, L1, s2, L2, th, post)
                      : orc_contralign_posterior(s1, L1, s2, L2, th, post);
  if (rc < 0) { free(post); return rc; }
  uint32_t n = 0;
  for (uint32_t i = 0; i != L1; ++i) {
    rowptr[i] = n;
    for (uint32_t j = 0; j != L2; ++j) {
      float p = post[(size_t)(L2 + 1) * (i + 1) + (j + 1)];
      if (p > th) { col[n] = j; val[n] = p; ++n; }
    }
  }
  rowptr[L1] = n;
  free(post);
  return (int)n;
}
