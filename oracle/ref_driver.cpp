// oracle/ref_driver.cpp -- TEST INFRASTRUCTURE ONLY (never linked into the product).
//
// Thin extern "C" wrapper that lets the Python tests call the *reference's own* code,
// compiled by oracle/Makefile straight from the sources where they lie under
// /root/reference/src (outputs only into oracle/_ref/, which is git-ignored).
// Nothing from the reference is copied here: this file only instantiates the reference's
// public classes and flattens their STL results into caller-owned C arrays.
//
// Reference entry points exercised:
//   ProbCons::calculate / CONTRAlign::calculate        src/align.cpp:60-106
//   PROBCONS::Probcons::ComputePosterior               src/probconsRNA/wrapper.cpp:101-131
//   CONTRALIGN::CONTRAlign<float>::ComputePosterior    src/contralign/wrapper.cpp:81-97
//   CONTRAFOLD::CONTRAfold<float>::ComputePosterior    src/contrafold/wrapper.cpp:181-200
//   SparseNussinov::decode (both overloads)            src/nussinov.cpp:207-392
//   SparseNeedlemanWunsch::initialize/decode           src/needleman_wunsch.cpp:198-422
//   Fasta::load                                        src/fa.cpp:37-87
//
// src/fold.cpp cannot be compiled here (it includes ViennaRNA headers, absent from this
// image), so the three static constants it defines (fold.cpp:56-58) that nussinov.cpp's
// make_brackets links against are defined below with the values stated there.
#include <string>
#include <vector>
#include <cstring>
#include <cstdint>

#include "align.h"
#include "fold.h"
#include "nussinov.h"
#include "needleman_wunsch.h"
#include "fa.h"

const uint Fold::Decoder::n_support_brackets = 4 + 26;                              // fold.cpp:56
const char* Fold::Decoder::left_brackets  = "([{<ABCDEFGHIJKLMNOPQRSTUVWXYZ";       // fold.cpp:57
const char* Fold::Decoder::right_brackets = ")]}>abcdefghijklmnopqrstuvwxyz";       // fold.cpp:58

static VVF to_vvf(const float* a, uint32_t r, uint32_t c) {
  VVF v(r, VF(c));
  for (uint32_t i = 0; i < r; ++i) std::memcpy(v[i].data(), a + (size_t)i * c, c * sizeof(float));
  return v;
}

extern "C" {

// dense posterior (L1+1)*(L2+1), values < th zeroed (wrapper.cpp:125-128)
int ref_probcons_posterior(const char* s1, const char* s2, float th, float* out) {
  static PROBCONS::Probcons pc;
  std::vector<float> p;
  pc.ComputePosterior(s1, s2, p, th);
  std::memcpy(out, p.data(), p.size() * sizeof(float));
  return (int)p.size();
}

int ref_contralign_posterior(const char* s1, const char* s2, float th, float* out) {
  static CONTRALIGN::CONTRAlign<float> ca;
  std::vector<float> p;
  ca.ComputePosterior(s1, s2, p, th);
  std::memcpy(out, p.data(), p.size() * sizeof(float));
  return (int)p.size();
}

// Align::Model adapters (align.cpp:60-106): MP as CSR. model 0 = ProbCons, 1 = CONTRAlign.
// rowptr[L1+1]; col/val capacity L1*L2.
int ref_align_calculate(int model, const char* s1, const char* s2, float th,
                        uint32_t* rowptr, uint32_t* col, float* val) {
  MP mp;
  if (model == 0) { ProbCons m(th); m.calculate(std::string(s1), std::string(s2), mp); }
  else            { CONTRAlign m(th); m.calculate(std::string(s1), std::string(s2), mp); }
  uint32_t n = 0;
  for (size_t i = 0; i < mp.size(); ++i) {
    rowptr[i] = n;
    for (auto& e : mp[i]) { col[n] = e.first; val[n] = e.second; ++n; }
  }
  rowptr[mp.size()] = n;
  return (int)n;
}

// CONTRAfold upper-triangular posterior, S=(L+1)(L+2)/2 floats (wrapper.cpp:181-200).
// constraint may be NULL (unconstrained).  A fresh object per call keeps SetConstraint
// state (wrapper.cpp:155-158) from leaking between calls.
int ref_contrafold_posterior(const char* seq, const char* constraint, float* out) {
  CONTRAFOLD::CONTRAfold<float> cf;   // canonical_only=true, max_bp_dist=0 (fold.cpp:170)
  std::vector<float> p;
  if (constraint) cf.SetConstraint(constraint);
  cf.ComputePosterior(seq, p);
  std::memcpy(out, p.data(), p.size() * sizeof(float));
  return (int)p.size();
}

float ref_contrafold_logz(const char* seq) {
  CONTRAFOLD::CONTRAfold<float> cf;
  cf.ComputeInside(seq);
  return cf.ComputeLogPartitionCoefficient();
}

// SparseNussinov::decode(w,p,q,ss)  (nussinov.cpp:207-298)
float ref_nussinov_decode(float th, float w, uint32_t L, const float* p, const float* q, uint32_t* ss) {
  SparseNussinov d(th);
  VU s;
  float r = d.decode(w, to_vvf(p, L, L), to_vvf(q, L, L), s);
  std::memcpy(ss, s.data(), L * sizeof(uint32_t));
  return r;
}

// SparseNussinov::decode(p,ss,str)  (nussinov.cpp:300-392); str must hold L+1 bytes
float ref_nussinov_decode_final(float th, uint32_t L, const float* p, uint32_t* ss, char* str) {
  SparseNussinov d(th);
  VU s; std::string b;
  float r = d.decode(to_vvf(p, L, L), s, b);
  std::memcpy(ss, s.data(), L * sizeof(uint32_t));
  std::memcpy(str, b.c_str(), L + 1);
  return r;
}

// SparseNeedlemanWunsch::initialize + decode (needleman_wunsch.cpp:198-422); q may be NULL.
// The envelope is private in the reference class, so it is observed only through decode.
float ref_nw_decode(float th, uint32_t L1, uint32_t L2, const float* p, const float* q, uint32_t* al) {
  SparseNeedlemanWunsch d(th);
  VVF P = to_vvf(p, L1, L2);
  d.initialize(P);
  VU a;
  float r = q ? d.decode(P, to_vvf(q, L1, L2), a) : d.decode(P, a);
  std::memcpy(al, a.data(), L1 * sizeof(uint32_t));
  return r;
}

// Fasta::load (fa.cpp:37-87): returns count; names/seqs concatenated with '\n' separators.
int ref_fasta_load(const char* file, char* names, size_t ncap, char* seqs, size_t scap) {
  std::vector<Fasta> fa;
  Fasta::load(fa, file);
  std::string n, s;
  for (auto& f : fa) { n += f.name(); n += '\n'; s += f.seq(); s += '\n'; }
  if (n.size() + 1 > ncap || s.size() + 1 > scap) return -1;
  std::memcpy(names, n.c_str(), n.size() + 1);
  std::memcpy(seqs, s.c_str(), s.size() + 1);
  return (int)fa.size();
}

}  // extern "C"
