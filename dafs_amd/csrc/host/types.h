// dafs_amd/csrc/host/types.h -- host-side containers of the DAFS interfaces, with the reference's
// names (reference src/typedefs.h:25-43, src/fa.h:28-68) so code written against the reference's
// plugin interfaces compiles unchanged against ours.
#pragma once
#include <string>
#include <utility>
#include <vector>

typedef unsigned int uint;
typedef std::vector<float> VF;
typedef std::vector<VF> VVF;
typedef std::vector<int> VI;
typedef std::vector<VI> VVI;
typedef std::vector<uint> VU;
typedef std::vector<VU> VVU;
typedef std::vector<std::pair<uint, float> > SV;  // sparse vector
typedef std::vector<SV> MP;                        // matching probabilities of one sequence pair
typedef std::vector<SV> BP;                        // base-pairing probabilities of one sequence
typedef std::vector<std::pair<uint, std::vector<bool> > > ALN;  // (sequence index, gap mask) per row

class Fasta {
 public:
  Fasta() {}
  Fasta(const std::string& name, const std::string& seq, const std::string& str = "") : name_(name), seq_(seq), str_(str) {}
  const std::string& name() const { return name_; }
  const std::string& seq() const { return seq_; }
  const std::string& str() const { return str_; }
  unsigned int size() const { return (unsigned int)seq_.size(); }
  // reads a FASTA file; throws std::system_error when it cannot be opened (reference src/fa.cpp:37-87)
  static unsigned int load(std::vector<Fasta>& data, const char* file);

 private:
  std::string name_, seq_, str_;
};
