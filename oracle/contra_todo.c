/* placeholder until the CONTRAlign restatement lands */
#include "oracle.h"
int orc_contralign_posterior(const char* s1, uint32_t L1, const char* s2, uint32_t L2, float th, float* out) { return -100; }
