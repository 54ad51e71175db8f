/* oracle/fasta.c -- TEST INFRASTRUCTURE (see oracle.h).
 * FASTA reader following Fasta::load, src/fa.cpp:37-87: a line whose first char is one of
 * "()[].?xle " is structure text (ignored here), anything else is sequence truncated at the
 * first non-alpha char; the header is the whole line after '>'.
 * PINNED against oracle/_ref's Fasta::load on the two example files. */
#include "oracle.h"
#include <ctype.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

int orc_fasta_load(const char* file, char*** names_out, char*** seqs_out) {
  FILE* f = fopen(file, "r");
  if (!f) return -1;
  int n = 0, cap = 16, have = 0;
  char** names = (char**)malloc(cap * sizeof(char*));
  char** seqs = (char**)malloc(cap * sizeof(char*));
  char* line = NULL;
  size_t lcap = 0;
  ssize_t len;
  char* name = NULL;
  char* seq = NULL;
  size_t slen = 0, scap = 0;
  while ((len = getline(&line, &lcap, f)) >= 0) {
    if (len > 0 && line[len - 1] == '\n') line[--len] = 0;
    if (line[0] == '>') {
      if (have && name[0]) {
        if (n == cap) { cap *= 2; names = (char**)realloc(names, cap * sizeof(char*)); seqs = (char**)realloc(seqs, cap * sizeof(char*)); }
        names[n] = name; seqs[n] = seq ? seq : strdup(""); n++;
      } else { free(name); free(seq); }
      name = strdup(line + 1);
      seq = NULL; slen = 0; scap = 0; have = 1;
      continue;
    }
    if (strchr("()[].?xle ", line[0]) == NULL || line[0] == 0) {
      /* note: strchr(s, '\0') != NULL in C, so an empty line counts as "structure" in the
       * reference (appends nothing); keep that by treating it as a no-op */
      if (line[0] == 0) continue;
      size_t i = 0;
      while (i < (size_t)len && isalpha((unsigned char)line[i])) i++;
      if (slen + i + 1 > scap) { scap = (slen + i + 1) * 2; seq = (char*)realloc(seq, scap); }
      memcpy(seq + slen, line, i);
      slen += i;
      seq[slen] = 0;
    }
  }
  if (have && name[0]) {
    if (n == cap) { cap *= 2; names = (char**)realloc(names, cap * sizeof(char*)); seqs = (char**)realloc(seqs, cap * sizeof(char*)); }
    names[n] = name; seqs[n] = seq ? seq : strdup(""); n++;
  } else { free(name); free(seq); }
  free(line);
  fclose(f);
  *names_out = names;
  *seqs_out = seqs;
  return n;
}

void orc_fasta_free(int n, char** names, char** seqs) {
  for (int i = 0; i < n; ++i) { free(names[i]); free(seqs[i]); }
  free(names); free(seqs);
}
