// dafs_amd/csrc/host/fasta.cpp -- FASTA reader with the behaviour of the reference's Fasta::load
// (reference src/fa.cpp:37-87): a record starts at '>' and its name is the rest of that line; a
// following line whose first character is one of "()[].?xle " is structure text, any other line
// is sequence, truncated at the first non-alphabetic character.
#include <cctype>
#include <cerrno>
#include <cstring>
#include <fstream>
#include <system_error>

#include "types.h"

unsigned int Fasta::load(std::vector<Fasta>& data, const char* file) {
  std::ifstream in(file);
  if (in.fail()) throw std::system_error(errno, std::system_category(), file);
  static const char kStructure[] = "()[].?xle ";
  std::string line, name, seq, str;
  bool open = false;
  auto flush = [&]() {
    if (open && !name.empty()) data.push_back(Fasta(name, seq, str));
  };
  while (std::getline(in, line)) {
    if (!line.empty() && line[0] == '>') {
      flush();
      name = line.substr(1);
      seq.clear();
      str.clear();
      open = true;
      continue;
    }
    // strchr also matches the terminating NUL, so an empty line counts as (empty) structure text
    const bool is_structure = line.empty() || std::strchr(kStructure, line[0]) != nullptr;
    size_t n = 0;
    if (is_structure) {
      while (n < line.size() && std::strchr(kStructure, line[n]) != nullptr && line[n] != '\0') ++n;
      str += line.substr(0, n);
    } else {
      while (n < line.size() && std::isalpha((unsigned char)line[n])) ++n;
      seq += line.substr(0, n);
    }
  }
  flush();
  return (unsigned int)data.size();
}
