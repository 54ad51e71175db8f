// dafs_amd/csrc/host/plugins.h -- the reference's four plugin interfaces (reference
// src/align.h:34-66, src/fold.h:30-61), and concrete classes that implement them on the GPU
// through the C ABI of libdafs_hip.so.  A DAFS maintainer drops these next to the reference's
// own ProbCons / CONTRAlign / CONTRAfold / SparseNussinov / SparseNeedlemanWunsch classes and
// selects them in DAFS::parse_options (INTEGRATION.md shows the three-line change).
//
// Error convention is the reference's: failures are thrown as `const char*`
// (caught in main, reference src/dafs.cpp:1893-1910).
#pragma once
#include <memory>
#include <string>
#include <vector>

#include "../../../include/dafs_hip.h"
#include "types.h"

namespace Align {
class Model {  // src/align.h:37-55
 public:
  Model(float th) : th_(th) {}
  virtual ~Model() {}
  virtual void calculate(const std::string& seq1, const std::string& seq2, MP& mp) = 0;
  virtual void calculate(const std::vector<Fasta>& fa, std::vector<std::vector<MP> >& mp);
  float threshold() const { return th_; }

 private:
  float th_;
};
class Decoder {  // src/align.h:57-65
 public:
  Decoder() {}
  virtual ~Decoder() {}
  virtual void initialize(const VVF& p) {}
  virtual float decode(const VVF& p, const VVF& q, VU& al) const = 0;
  virtual float decode(const VVF& p, VU& al) const = 0;
};
}  // namespace Align

namespace Fold {
class Model {  // src/fold.h:33-45
 public:
  Model(float th) : th_(th) {}
  virtual ~Model() {}
  virtual void calculate(const std::string& seq, BP& bp) = 0;
  virtual void calculate(const std::string& seq, const std::string& str, BP& bp) = 0;
  virtual void calculate(const std::vector<Fasta>& fa, std::vector<BP>& bp);
  float threshold() const { return th_; }

 private:
  float th_;
};
class Decoder {  // src/fold.h:47-60
 public:
  Decoder() {}
  virtual ~Decoder() {}
  virtual float decode(float w, const VVF& p, const VVF& q, VU& ss) = 0;
  virtual float decode(const VVF& p, VU& ss, std::string& str) = 0;
  virtual void make_brackets(const VU& ss, std::string& str) const = 0;
};
}  // namespace Fold

// One GPU context shared by the plugin objects of a process.
class HipContext {
 public:
  explicit HipContext(int device = 0);
  ~HipContext();
  dafs_hip_ctx* get() const { return ctx_; }
  static void check(int rc);  // throws dafs_hip_strerror(rc) as const char* when rc != 0

 private:
  dafs_hip_ctx* ctx_;
};

// -a ProbCons / -a CONTRAlign on the GPU.  The batch overload is the hook the reference calls
// (src/dafs.cpp:1796): all N(N-1)/2 pairs in one launch.
class HipAlignModel : public Align::Model {
 public:
  HipAlignModel(std::shared_ptr<HipContext> ctx, int model, float th) : Align::Model(th), ctx_(ctx), model_(model) {}
  void calculate(const std::string& seq1, const std::string& seq2, MP& mp);
  void calculate(const std::vector<Fasta>& fa, std::vector<std::vector<MP> >& mp);

 private:
  std::shared_ptr<HipContext> ctx_;
  int model_;
};

// -s CONTRAfold on the GPU
class HipCONTRAfold : public Fold::Model {
 public:
  HipCONTRAfold(std::shared_ptr<HipContext> ctx, float th) : Fold::Model(th), ctx_(ctx) {}
  void calculate(const std::string& seq, BP& bp);
  void calculate(const std::string& seq, const std::string& str, BP& bp);
  void calculate(const std::vector<Fasta>& fa, std::vector<BP>& bp);

 private:
  std::shared_ptr<HipContext> ctx_;
};

class HipSparseNussinov : public Fold::Decoder {
 public:
  HipSparseNussinov(std::shared_ptr<HipContext> ctx, float th) : ctx_(ctx), th_(th) {}
  float decode(float w, const VVF& p, const VVF& q, VU& ss);
  float decode(const VVF& p, VU& ss, std::string& str);
  void make_brackets(const VU& ss, std::string& str) const;

 private:
  std::shared_ptr<HipContext> ctx_;
  float th_;
};

// the dense decoder classes of the reference (src/nussinov.h:25-35, src/needleman_wunsch.h:26-36)
class HipNussinov : public Fold::Decoder {
 public:
  HipNussinov(std::shared_ptr<HipContext> ctx, float th) : ctx_(ctx), th_(th) {}
  float decode(float w, const VVF& p, const VVF& q, VU& ss);
  float decode(const VVF& p, VU& ss, std::string& str);
  void make_brackets(const VU& ss, std::string& str) const;

 private:
  std::shared_ptr<HipContext> ctx_;
  float th_;
};

class HipNeedlemanWunsch : public Align::Decoder {
 public:
  HipNeedlemanWunsch(std::shared_ptr<HipContext> ctx, float th) : ctx_(ctx), th_(th) {}
  void initialize(const VVF&) {}
  float decode(const VVF& p, const VVF& q, VU& al) const;
  float decode(const VVF& p, VU& al) const;

 private:
  std::shared_ptr<HipContext> ctx_;
  float th_;
};

class HipSparseNeedlemanWunsch : public Align::Decoder {
 public:
  HipSparseNeedlemanWunsch(std::shared_ptr<HipContext> ctx, float th) : ctx_(ctx), th_(th) {}
  void initialize(const VVF& p);
  float decode(const VVF& p, const VVF& q, VU& al) const;
  float decode(const VVF& p, VU& al) const;

 private:
  std::shared_ptr<HipContext> ctx_;
  float th_;
  std::vector<uint32_t> env_;
};
