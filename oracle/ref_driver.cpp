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
//   Nussinov::decode (both overloads, dense)           src/nussinov.cpp:32-204
//   NeedlemanWunsch::decode (both overloads, dense)    src/needleman_wunsch.cpp:28-196
//   AUXAlign::calculate (--align-aux reader)           src/align.cpp:204-246
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

// Nussinov::decode(w,p,q,ss) / decode(p,ss,str): the dense decoder (nussinov.cpp:32-204); q may be NULL (final overload)
float ref_nussinov_dense_decode(float th, float w, uint32_t L, const float* p, const float* q, uint32_t* ss) {
  Nussinov d(th);
  VU s;
  float r;
  if (q) r = d.decode(w, to_vvf(p, L, L), to_vvf(q, L, L), s);
  else { std::string b; r = d.decode(to_vvf(p, L, L), s, b); }
  std::memcpy(ss, s.data(), L * sizeof(uint32_t));
  return r;
}

// NeedlemanWunsch::decode (needleman_wunsch.cpp:28-196); q may be NULL
float ref_nw_dense_decode(float th, uint32_t L1, uint32_t L2, const float* p, const float* q, uint32_t* al) {
  NeedlemanWunsch d(th);
  VVF P = to_vvf(p, L1, L2);
  d.initialize(P);
  VU a;
  float r = q ? d.decode(P, to_vvf(q, L1, L2), a) : d.decode(P, a);
  std::memcpy(al, a.data(), L1 * sizeof(uint32_t));
  return r;
}

// AUXAlign::calculate(fa, mp) (align.cpp:204-246): the reference's own reader of the --align-aux format.
// seqs: nseq strings (only their lengths matter to the reader).  Output, for every pair x < y in row-major order:
// nnz[p], then len_x+1 row pointers (relative to the pair), then the entries -- the layout of dafs_hip_set_mp.
// Returns the total number of entries, or -1 when a capacity is too small.
long ref_auxalign_load(const char* file, int nseq, const char* const* seqs, uint32_t* nnz, uint32_t* rowptr, size_t rp_cap,
                       uint32_t* col, float* val, size_t ent_cap) {
  std::vector<Fasta> fa;
  for (int i = 0; i < nseq; ++i) fa.push_back(Fasta("s", seqs[i]));
  AUXAlign m(file, 0.0f);
  std::vector<std::vector<MP> > mp;
  m.calculate(fa, mp);
  size_t rp = 0, e = 0, pidx = 0;
  for (int x = 0; x < nseq; ++x)
    for (int y = x + 1; y < nseq; ++y, ++pidx) {
      const MP& m2 = mp[x][y];
      const size_t L1 = fa[x].size();
      if (rp + L1 + 1 > rp_cap) return -1;
      uint32_t n = 0;
      for (size_t i = 0; i < L1; ++i) {
        rowptr[rp + i] = n;
        if (i < m2.size())
          for (auto& en : m2[i]) {
            if (e >= ent_cap) return -1;
            col[e] = en.first; val[e] = en.second; ++e; ++n;
          }
      }
      rowptr[rp + L1] = n;
      nnz[pidx] = n;
      rp += L1 + 1;
    }
  return (long)e;
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
