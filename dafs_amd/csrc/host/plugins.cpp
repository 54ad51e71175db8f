// dafs_amd/csrc/host/plugins.cpp -- see plugins.h
#include "plugins.h"

#include <cstring>

void HipContext::check(int rc) {
  if (rc != DAFS_HIP_OK) throw dafs_hip_strerror(rc);
}
HipContext::HipContext(int device) : ctx_(nullptr) { check(dafs_hip_create(device, &ctx_)); }
HipContext::~HipContext() { dafs_hip_destroy(ctx_); }

// ---- base-class batch loops, as the reference defines them (src/align.cpp:35-52, src/fold.cpp:60-68)
void Align::Model::calculate(const std::vector<Fasta>& fa, std::vector<std::vector<MP> >& mp) {
  const uint N = (uint)fa.size();
  mp.assign(N, std::vector<MP>(N));
  for (uint i = 0; i != N; ++i) {
    mp[i][i].resize(fa[i].size());
    for (uint x = 0; x != fa[i].size(); ++x) mp[i][i][x].push_back(std::make_pair(x, 1.0f));
    for (uint j = i + 1; j != N; ++j) this->calculate(fa[i].seq(), fa[j].seq(), mp[i][j]);
  }
}
void Fold::Model::calculate(const std::vector<Fasta>& fa, std::vector<BP>& bp) {
  bp.resize(fa.size());
  for (uint i = 0; i != fa.size(); ++i) this->calculate(fa[i].seq(), bp[i]);
}

static void flatten(const VVF& m, std::vector<float>& out) {
  const size_t R = m.size(), C = R ? m[0].size() : 0;
  out.resize(R * C);
  for (size_t i = 0; i < R; ++i) std::memcpy(out.data() + i * C, m[i].data(), C * sizeof(float));
}

// ---- alignment model
static void fetch_pairs(dafs_hip_ctx* c, const std::vector<uint32_t>& lens, std::vector<std::vector<MP> >& mp) {
  uint64_t np = 0, nnz = 0, nrp = 0;
  HipContext::check(dafs_hip_align_result_size(c, &np, &nnz, &nrp));
  std::vector<uint32_t> px(np), py(np), cnt(np), rowptr(nrp), col(2 * nnz);
  std::vector<float> val(2 * nnz);
  HipContext::check(dafs_hip_align_fetch(c, px.data(), py.data(), nullptr, cnt.data(), rowptr.data(), col.data(), val.data()));
  size_t r = 0, e = 0;
  for (uint64_t p = 0; p < np; ++p) {
    const uint32_t L1 = lens[px[p]], L2 = lens[py[p]];
    MP& m = mp[px[p]][py[p]];
    m.assign(L1, SV());
    for (uint32_t i = 0; i < L1; ++i)
      for (uint32_t k = rowptr[r + i]; k < rowptr[r + i + 1]; ++k) m[i].push_back(std::make_pair(col[e + k], val[e + k]));
    r += (size_t)L1 + 1 + L2 + 1;
    e += 2 * (size_t)cnt[p];
  }
}

void HipAlignModel::calculate(const std::vector<Fasta>& fa, std::vector<std::vector<MP> >& mp) {
  const uint N = (uint)fa.size();
  mp.assign(N, std::vector<MP>(N));
  std::vector<const char*> seqs(N);
  std::vector<uint32_t> lens(N);
  for (uint i = 0; i < N; ++i) { seqs[i] = fa[i].seq().c_str(); lens[i] = fa[i].size(); }
  HipContext::check(dafs_hip_set_sequences(ctx_->get(), N, seqs.data(), lens.data()));
  if (N > 1) {
    HipContext::check(dafs_hip_align_posteriors(ctx_->get(), model_, threshold(), 0, 0));
    fetch_pairs(ctx_->get(), lens, mp);
  }
  for (uint i = 0; i != N; ++i) {  // identity on the diagonal, src/align.cpp:42-44
    mp[i][i].assign(fa[i].size(), SV());
    for (uint x = 0; x != fa[i].size(); ++x) mp[i][i][x].push_back(std::make_pair(x, 1.0f));
  }
}

void HipAlignModel::calculate(const std::string& seq1, const std::string& seq2, MP& mp) {
  std::vector<Fasta> fa;
  fa.push_back(Fasta("1", seq1));
  fa.push_back(Fasta("2", seq2));
  std::vector<std::vector<MP> > all;
  calculate(fa, all);
  mp.swap(all[0][1]);
}

// ---- folding model
static void triangular_to_bp(const std::vector<float>& post, uint L, float th, BP& bp) {  // src/fold.cpp:181-188
  bp.assign(L, SV());
  size_t k = 0;
  for (uint i = 0; i != L + 1; ++i)
    for (uint j = i; j != L + 1; ++j, ++k)
      if (i != 0 && post[k] > th) bp[i - 1].push_back(std::make_pair(j - 1, post[k]));
}

void HipCONTRAfold::calculate(const std::string& seq, BP& bp) {
  std::vector<float> post((size_t)(seq.size() + 1) * (seq.size() + 2) / 2);
  HipContext::check(dafs_hip_fold_posterior_dense(ctx_->get(), seq.c_str(), (uint32_t)seq.size(), nullptr, post.data(), nullptr));
  triangular_to_bp(post, (uint)seq.size(), threshold(), bp);
}
void HipCONTRAfold::calculate(const std::string& seq, const std::string& str, BP& bp) {
  std::vector<float> post((size_t)(seq.size() + 1) * (seq.size() + 2) / 2);
  HipContext::check(dafs_hip_fold_posterior_dense(ctx_->get(), seq.c_str(), (uint32_t)seq.size(), str.c_str(), post.data(), nullptr));
  triangular_to_bp(post, (uint)seq.size(), threshold(), bp);
}
void HipCONTRAfold::calculate(const std::vector<Fasta>& fa, std::vector<BP>& bp) {
  const uint N = (uint)fa.size();
  std::vector<const char*> seqs(N);
  std::vector<uint32_t> lens(N);
  for (uint i = 0; i < N; ++i) { seqs[i] = fa[i].seq().c_str(); lens[i] = fa[i].size(); }
  dafs_hip_ctx* c = ctx_->get();
  HipContext::check(dafs_hip_set_sequences(c, N, seqs.data(), lens.data()));
  HipContext::check(dafs_hip_fold_posteriors(c, DAFS_FOLD_CONTRAFOLD, threshold()));
  uint64_t nnz = 0, nrp = 0;
  HipContext::check(dafs_hip_bp_result_size(c, 0, &nnz, &nrp));
  std::vector<uint32_t> rowptr(nrp), col(nnz);
  std::vector<float> val(nnz);
  HipContext::check(dafs_hip_bp_fetch(c, 0, rowptr.data(), col.data(), val.data()));
  bp.assign(N, BP());
  size_t r = 0, e = 0;
  for (uint x = 0; x < N; ++x) {
    bp[x].assign(lens[x], SV());
    for (uint32_t i = 0; i < lens[x]; ++i)
      for (uint32_t k = rowptr[r + i]; k < rowptr[r + i + 1]; ++k) bp[x][i].push_back(std::make_pair(col[e + k], val[e + k]));
    e += rowptr[r + lens[x]];
    r += (size_t)lens[x] + 1;
  }
}

// ---- decoders
float HipSparseNussinov::decode(float w, const VVF& p, const VVF& q, VU& ss) {
  std::vector<float> fp, fq;
  flatten(p, fp);
  flatten(q, fq);
  ss.assign(p.size(), -1u);
  float score = 0;
  HipContext::check(dafs_hip_nussinov_decode(ctx_->get(), th_, w, (uint32_t)p.size(), fp.data(), fq.data(), ss.data(), &score));
  return score;
}
float HipSparseNussinov::decode(const VVF& p, VU& ss, std::string& str) {
  std::vector<float> fp;
  flatten(p, fp);
  ss.assign(p.size(), -1u);
  float score = 0;
  HipContext::check(dafs_hip_nussinov_decode(ctx_->get(), th_, 0.0f, (uint32_t)p.size(), fp.data(), nullptr, ss.data(), &score));
  make_brackets(ss, str);
  return score;
}
void HipSparseNussinov::make_brackets(const VU& ss, std::string& str) const {
  std::vector<char> buf(ss.size() + 1);
  dafs_hip_make_brackets((uint32_t)ss.size(), ss.data(), buf.data());
  str.assign(buf.data());
}

float HipNussinov::decode(float w, const VVF& p, const VVF& q, VU& ss) {
  std::vector<float> fp, fq;
  flatten(p, fp);
  flatten(q, fq);
  ss.assign(p.size(), -1u);
  float score = 0;
  HipContext::check(dafs_hip_nussinov_decode_dense(ctx_->get(), th_, w, (uint32_t)p.size(), fp.data(), fq.data(), ss.data(), &score));
  return score;
}
float HipNussinov::decode(const VVF& p, VU& ss, std::string& str) {
  std::vector<float> fp;
  flatten(p, fp);
  ss.assign(p.size(), -1u);
  float score = 0;
  HipContext::check(dafs_hip_nussinov_decode_dense(ctx_->get(), th_, 0.0f, (uint32_t)p.size(), fp.data(), nullptr, ss.data(), &score));
  make_brackets(ss, str);
  return score;
}
void HipNussinov::make_brackets(const VU& ss, std::string& str) const {
  std::vector<char> buf(ss.size() + 1);
  dafs_hip_make_brackets((uint32_t)ss.size(), ss.data(), buf.data());
  str.assign(buf.data());
}
float HipNeedlemanWunsch::decode(const VVF& p, const VVF& q, VU& al) const {
  std::vector<float> fp, fq;
  flatten(p, fp);
  flatten(q, fq);
  al.assign(p.size(), -1u);
  float score = 0;
  HipContext::check(dafs_hip_nw_decode_dense(ctx_->get(), th_, (uint32_t)p.size(), (uint32_t)p[0].size(), fp.data(), fq.data(), al.data(), &score));
  return score;
}
float HipNeedlemanWunsch::decode(const VVF& p, VU& al) const {
  std::vector<float> fp;
  flatten(p, fp);
  al.assign(p.size(), -1u);
  float score = 0;
  HipContext::check(dafs_hip_nw_decode_dense(ctx_->get(), th_, (uint32_t)p.size(), (uint32_t)p[0].size(), fp.data(), nullptr, al.data(), &score));
  return score;
}

void HipSparseNeedlemanWunsch::initialize(const VVF& p) {
  std::vector<float> fp;
  flatten(p, fp);
  env_.assign(2 * (p.size() + 1), 0);
  HipContext::check(dafs_hip_nw_envelope(ctx_->get(), th_, (uint32_t)p.size(), (uint32_t)p[0].size(), fp.data(), env_.data()));
}
float HipSparseNeedlemanWunsch::decode(const VVF& p, const VVF& q, VU& al) const {
  std::vector<float> fp, fq;
  flatten(p, fp);
  flatten(q, fq);
  al.assign(p.size(), -1u);
  float score = 0;
  HipContext::check(dafs_hip_nw_decode(ctx_->get(), th_, (uint32_t)p.size(), (uint32_t)p[0].size(), fp.data(), fq.data(), env_.data(), al.data(), &score));
  return score;
}
float HipSparseNeedlemanWunsch::decode(const VVF& p, VU& al) const {
  std::vector<float> fp;
  flatten(p, fp);
  al.assign(p.size(), -1u);
  float score = 0;
  HipContext::check(dafs_hip_nw_decode(ctx_->get(), th_, (uint32_t)p.size(), (uint32_t)p[0].size(), fp.data(), nullptr, env_.data(), al.data(), &score));
  return score;
}
