/* placeholders until the CONTRAlign / CONTRAfold restatements land */
#include "oracle.h"
int orc_contralign_posterior(const char* s1, uint32_t L1, const char* s2, uint32_t L2, float th, float* out) { return -100; }
int orc_contrafold_posterior(const char* seq, uint32_t L, const char* constraint, float* out) { return -100; }
float orc_contrafold_logz(const char* seq, uint32_t L) { return 0.0f / 0.0f; }
int orc_fold_calculate(const char* seq, uint32_t L, const char* constraint, float th, uint32_t* rowptr, uint32_t* col, float* val) { return -100; }
