// dafs_amd/csrc/host/cli_main.cpp -- the `dafs [OPTIONS] FASTA` command line on top of libdafs_hip.so.
//
// Same flags, defaults and output format as the reference program (reference src/dafs.cpp:1603-1779
// for the options, :1781-1889 for the run, :495-511 and :1584-1601 for the output), with every
// numeric stage executed on the GPU through the C ABI: base-pairing posteriors, all-pairs matching
// posteriors, similarity, both consistency transforms, and per guide-tree node the averaging +
// dual-decomposition solve.  The host keeps what is inherently serial and tiny: option parsing,
// the guide tree, the alignment bookkeeping (project_alignment) and printing.
//
// Differences from the reference, all forced by what its tree does not contain:
//   -s Boltzmann / -s Vienna and the RNAalifold term need ViennaRNA arithmetic: not available.
//      The default fold model here is CONTRAfold; asking for the others is an error.
//   --fold-decoder IPknot / --ipknot / -m 0 need an ILP solver: not available.
// One addition: --devices a,b,... runs phase 1 (:1787-1827) as one process per listed GPU (dafs_hip_phase1_sharded, the
// shards all-gathered by RCCL); the processes are forked before anything touches a GPU, and the first one goes on alone.
#include <dlfcn.h>
#include <pthread.h>
#include <signal.h>
#include <sys/mman.h>
#include <sys/prctl.h>
#include <sys/wait.h>
#include <unistd.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <map>
#include <queue>
#include <sstream>
#include <string>
#include <system_error>
#include <vector>

#ifndef __HIP_PLATFORM_AMD__
#define __HIP_PLATFORM_AMD__ 1
#endif
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>  // types and prototypes only: librccl is loaded when --devices asks for it (dlopen)

#include "../../../include/dafs_hip.h"
#include "types.h"

#define DAFS_VERSION "0.0.4-hip"
static const float kCutoff = 0.01f;  // reference CUTOFF (src/dafs.cpp:65)

namespace {

void check(int rc) {
  if (rc != DAFS_HIP_OK) throw dafs_hip_strerror(rc);
}

struct Options {
  int refinement = 0;
  float w = 4.0f, eta = 0.5f;
  int max_iter = 600;
  float fourway = 0.0f;
  int verbose = 0;
  std::string align_model = "ProbCons";
  float align_pct = 0.25f, align_th = 0.01f;
  std::string align_aux, fold_aux, save_align_aux, save_fold_aux;
  std::string fold_model = "CONTRAfold";
  bool fold_model_given = false;
  std::string fold_decoder = "Nussinov";
  float fold_pct = 0.25f;
  std::vector<float> fold_th{0.2f}, fold_th1;
  bool no_alifold = false, ipknot = false, bp_update = false, bp_update1 = false;
  int device = 0;
  std::vector<int> devices;  // --devices: one process per entry for phase 1
  std::string input;
};

const char* kHelp =
    "DAFS: dual decomposition for simultaneous aligning and folding RNA sequences (MI355X build).\n"
    "Usage:\n  dafs [OPTION...] FILE\n\n"
    "  -h, --help            Print usage\n"
    "      --version         Print version\n"
    "  -r, --refinement N    The number of iteration of the iterative refinment (default: 0)\n"
    "  -w, --weight arg      Weight of the expected accuracy score for secondary structures (default: 4.0)\n"
    "      --eta arg         Initial step width for the subgradient optimization (default: 0.5)\n"
    "  -m, --max-iter T      The maximum number of iteration of the subgradient optimization (default: 600)\n"
    "  -f, --fourway-pct arg Weight of four-way PCT (default: 0.0)\n"
    "  -v, --verbose arg     The level of verbose outputs (default: 0)\n"
    "      --device N        HIP device index (default: 0)\n"
    "      --devices A,B,... One process per listed HIP device for the posteriors and the consistency transform\n"
    "                        (shards exchanged over RCCL); the progressive alignment runs on the first\n"
    "\n Aligning options:\n"
    "  -a, --align-model arg Alignment model (value=CONTRAlign, ProbCons) (default: ProbCons)\n"
    "  -p, --align-pct arg   Weight of PCT for matching probabilities (default: 0.25)\n"
    "  -u, --align-th arg    Threshold for matching probabilities (default: 0.01)\n"
    "      --save-align-aux FILENAME  Write matching probability matrices in the --align-aux format\n"
    "\n Folding options:\n"
    "  -s, --fold-model arg  Folding model (value=CONTRAfold; Boltzmann and Vienna need ViennaRNA and are\n"
    "                        not available in this build) (default: CONTRAfold)\n"
    "      --fold-decoder arg Decoder for common secondary structure prediction (value=Nussinov) (default: Nussinov)\n"
    "  -q, --fold-pct arg    Weight of PCT for base-pairing probabilities (default: 0.25)\n"
    "  -t, --fold-th arg     Threshold for base-pairing probabilities (default: 0.2)\n"
    "  -g, --gamma arg       Specify the threshold for base-pairing probabilities by 1/(gamma+1)\n"
    "      --no-alifold      No use of RNAalifold (always the case in this build)\n"
    "  -T, --fold-th1 arg    Threshold for base-pairing probabilities of the conclusive common secondary structures\n"
    "  -G, --gamma1 arg      ... specified by 1/(gamma+1)\n"
    "      --bp-update       Re-estimate the base-pairing matrices of the last alignment step under the predicted structure\n"
    "      --bp-update1      The same for the conclusive common secondary structure\n"
    "      --fold-aux FILENAME        Load base-pairing probability matrices from FILENAME\n"
    "      --save-fold-aux FILENAME   Write base-pairing probability matrices in the --fold-aux format\n";

std::vector<float> parse_floats(const std::string& s) {  // cxxopts vector<float>: comma separated
  std::vector<float> v;
  std::stringstream ss(s);
  std::string item;
  while (std::getline(ss, item, ',')) v.push_back(std::stof(item));
  return v;
}

Options parse(int argc, char** argv) {
  Options o;
  // long name -> (short char, takes value)
  const std::map<std::string, std::pair<char, bool> > spec = {
      {"help", {'h', false}}, {"version", {0, false}}, {"refinement", {'r', true}}, {"weight", {'w', true}}, {"eta", {0, true}},
      {"max-iter", {'m', true}}, {"fourway-pct", {'f', true}}, {"verbose", {'v', true}}, {"align-model", {'a', true}},
      {"align-pct", {'p', true}}, {"align-th", {'u', true}}, {"align-aux", {0, true}}, {"fold-model", {'s', true}},
      {"fold-decoder", {0, true}}, {"fold-pct", {'q', true}}, {"fold-th", {'t', true}}, {"gamma", {'g', true}},
      {"no-alifold", {0, false}}, {"fold-th1", {'T', true}}, {"gamma1", {'G', true}}, {"ipknot", {0, false}},
      {"bp-update", {0, false}}, {"bp-update1", {0, false}}, {"fold-aux", {0, true}}, {"save-align-aux", {0, true}},
      {"save-fold-aux", {0, true}}, {"device", {0, true}}, {"devices", {0, true}}, {"input", {0, true}}};
  std::map<char, std::string> shorts;
  for (const auto& kv : spec)
    if (kv.second.first) shorts[kv.second.first] = kv.first;
  std::vector<float> gamma, gamma1;
  bool th_given = false, th1_given = false;
  for (int i = 1; i < argc; ++i) {
    std::string arg = argv[i], name, value;
    bool have_value = false;
    if (arg.size() > 2 && arg[0] == '-' && arg[1] == '-') {
      const size_t eq = arg.find('=');
      name = arg.substr(2, eq == std::string::npos ? std::string::npos : eq - 2);
      if (eq != std::string::npos) { value = arg.substr(eq + 1); have_value = true; }
    } else if (arg.size() >= 2 && arg[0] == '-' && !(arg[1] >= '0' && arg[1] <= '9')) {
      if (!shorts.count(arg[1])) throw std::string("Unknown option: ") + arg;
      name = shorts[arg[1]];
      if (arg.size() > 2) { value = arg.substr(2); have_value = true; }
    } else {
      o.input = arg;
      continue;
    }
    const auto it = spec.find(name);
    if (it == spec.end()) throw std::string("Unknown option: ") + arg;
    if (it->second.second && !have_value) {
      if (i + 1 >= argc) throw std::string("Option requires a value: ") + arg;
      value = argv[++i];
    }
    if (name == "help") { std::cout << kHelp << std::endl; exit(0); }
    else if (name == "version") { std::cout << "DAFS version " << DAFS_VERSION << std::endl; exit(0); }
    else if (name == "refinement") o.refinement = std::stoi(value);
    else if (name == "weight") o.w = std::stof(value);
    else if (name == "eta") o.eta = std::stof(value);
    else if (name == "max-iter") o.max_iter = std::stoi(value);
    else if (name == "fourway-pct") o.fourway = std::stof(value);
    else if (name == "verbose") o.verbose = std::stoi(value);
    else if (name == "align-model") o.align_model = value;
    else if (name == "align-pct") o.align_pct = std::stof(value);
    else if (name == "align-th") o.align_th = std::stof(value);
    else if (name == "align-aux") o.align_aux = value;
    else if (name == "fold-model") { o.fold_model = value; o.fold_model_given = true; }
    else if (name == "fold-decoder") o.fold_decoder = value;
    else if (name == "fold-pct") o.fold_pct = std::stof(value);
    else if (name == "fold-th") { o.fold_th = parse_floats(value); th_given = true; }
    else if (name == "gamma") gamma = parse_floats(value);
    else if (name == "no-alifold") o.no_alifold = true;
    else if (name == "fold-th1") { o.fold_th1 = parse_floats(value); th1_given = true; }
    else if (name == "gamma1") gamma1 = parse_floats(value);
    else if (name == "ipknot") o.ipknot = true;
    else if (name == "bp-update") o.bp_update = true;
    else if (name == "bp-update1") o.bp_update1 = true;
    else if (name == "fold-aux") o.fold_aux = value;
    else if (name == "save-align-aux") o.save_align_aux = value;
    else if (name == "save-fold-aux") o.save_fold_aux = value;
    else if (name == "device") o.device = std::stoi(value);
    else if (name == "devices") {
      for (float d : parse_floats(value)) o.devices.push_back((int)d);
      if (o.devices.empty()) throw std::string("--devices needs at least one device index");
    }
    else if (name == "input") o.input = value;
  }
  // thresholds, reference src/dafs.cpp:1709-1750
  if (!th_given && !gamma.empty()) {
    o.fold_th = gamma;
    for (float& t : o.fold_th) t = 1.0 / (1.0 + t);
  }
  if (!th1_given) {
    if (!gamma1.empty()) {
      o.fold_th1 = gamma1;
      for (float& t : o.fold_th1) t = 1.0 / (1.0 + t);
    } else {
      o.fold_th1 = o.fold_th;
    }
  }
  if (o.input.empty()) { std::cout << kHelp << std::endl; exit(0); }
  return o;
}

// ---------------------------------------------------------------------------------------------
typedef std::pair<float, std::pair<uint, uint> > node_t;

// DAFS::build_tree, reference src/dafs.cpp:446-492 (dafs_host_build_tree in the library; host code)
std::vector<node_t> build_tree(const std::vector<float>& sim, uint n0) {
  std::vector<float> score(2 * n0 - 1);
  std::vector<int32_t> left(2 * n0 - 1), right(2 * n0 - 1);
  check(dafs_host_build_tree(n0, sim.data(), score.data(), left.data(), right.data()));
  std::vector<node_t> tree(2 * n0 - 1);
  for (uint i = 0; i < 2 * n0 - 1; ++i) tree[i] = std::make_pair(score[i], std::make_pair((uint)left[i], (uint)right[i]));
  return tree;
}

void print_tree(std::ostream& os, const std::vector<node_t>& tree, const std::vector<Fasta>& fa, int i) {  // :495-511
  if (tree[i].second.first == -1u) { os << fa[i].name(); return; }
  os << "[ " << tree[i].first << " ";
  print_tree(os, tree, fa, tree[i].second.first);
  os << " ";
  print_tree(os, tree, fa, tree[i].second.second);
  os << " ]";
}

// DAFS::project_alignment, :766-825
void project_alignment(ALN& aln, const ALN& a1, const ALN& a2, const VU& z) {
  const uint L1 = (uint)a1[0].second.size(), L2 = (uint)a2[0].second.size();
  std::vector<int> c1, c2;  // per merged column: source column of aln1 / aln2, or -1
  uint k = 0;
  for (uint i = 0; i != L1; ++i) {
    if (z[i] != -1u) {
      while (k < z[i]) { c1.push_back(-1); c2.push_back((int)k++); }
      c1.push_back((int)i);
      c2.push_back((int)k++);
    } else {
      c1.push_back((int)i);
      c2.push_back(-1);
    }
  }
  while (k < L2) { c1.push_back(-1); c2.push_back((int)k++); }
  const size_t L = c1.size();
  aln.clear();
  for (const auto& row : a1) {
    std::vector<bool> m(L, false);
    for (size_t c = 0; c < L; ++c) m[c] = c1[c] >= 0 && row.second[c1[c]];
    aln.push_back(std::make_pair(row.first, m));
  }
  for (const auto& row : a2) {
    std::vector<bool> m(L, false);
    for (size_t c = 0; c < L; ++c) m[c] = c2[c] >= 0 && row.second[c2[c]];
    aln.push_back(std::make_pair(row.first, m));
  }
}

struct NodeJob {  // flattened child alignments of one node, kept alive across the C call
  std::vector<uint32_t> s1, s2;
  std::vector<uint8_t> m1, m2;
  std::vector<float> px, py;  // --bp-update: re-estimated base-pairing matrices
  VU x, y, z;
};

// the `if (use_bp_update_)` blocks of align_alignments(ss, ...), :919-934, and of DAFS::run, :1863-1869: decode the averaged
// matrix, re-estimate it under that structure (update_basepairing_probability, :609-712)
void updated_bp(dafs_hip_ctx* ctx, uint32_t n, uint32_t len, const std::vector<uint32_t>& seq, const std::vector<uint8_t>& mask, float th,
                std::vector<float>& p) {
  VU ss(len);
  check(dafs_hip_consensus_structure(ctx, n, len, seq.data(), mask.data(), th, ss.data(), nullptr, nullptr));
  p.resize((size_t)len * len);
  check(dafs_hip_update_basepairing(ctx, n, len, seq.data(), mask.data(), ss.data(), p.data()));
}
void flatten(const ALN& a, std::vector<uint32_t>& s, std::vector<uint8_t>& m) {
  const size_t L = a[0].second.size();
  s.resize(a.size());
  m.resize(a.size() * L);
  for (size_t r = 0; r < a.size(); ++r) {
    s[r] = a[r].first;
    for (size_t c = 0; c < L; ++c) m[r * L + c] = a[r].second[c] ? 1 : 0;
  }
}

// align_alignments for a batch of independent (aln1, aln2) pairs: :896-981
std::vector<float> solve_batch(dafs_hip_ctx* ctx, const dafs_dd_params& prm, const std::vector<const ALN*>& a1,
                               const std::vector<const ALN*>& a2, std::vector<ALN>& out, int verbose, bool bp_update) {
  const size_t nb = a1.size();
  std::vector<NodeJob> jobs(nb);
  std::vector<dafs_node_input> in(nb);
  std::vector<dafs_node_output> res(nb);
  for (size_t b = 0; b < nb; ++b) {
    NodeJob& j = jobs[b];
    flatten(*a1[b], j.s1, j.m1);
    flatten(*a2[b], j.s2, j.m2);
    in[b].n1 = (uint32_t)a1[b]->size(); in[b].n2 = (uint32_t)a2[b]->size();
    in[b].len1 = (uint32_t)(*a1[b])[0].second.size(); in[b].len2 = (uint32_t)(*a2[b])[0].second.size();
    in[b].seq1 = j.s1.data(); in[b].seq2 = j.s2.data(); in[b].mask1 = j.m1.data(); in[b].mask2 = j.m2.data();
    j.x.resize(in[b].len1); j.y.resize(in[b].len2); j.z.resize(in[b].len1);
    res[b].x = j.x.data(); res[b].y = j.y.data(); res[b].z = j.z.data();
  }
  if (bp_update)
    for (size_t b = 0; b < nb; ++b) {  // refine() goes through align_alignments(ss, ...) too
      updated_bp(ctx, in[b].n1, in[b].len1, jobs[b].s1, jobs[b].m1, prm.th_s, jobs[b].px);
      updated_bp(ctx, in[b].n2, in[b].len2, jobs[b].s2, jobs[b].m2, prm.th_s, jobs[b].py);
      in[b].p_x = jobs[b].px.data(); in[b].p_y = jobs[b].py.data();
    }
  check(dafs_hip_solve_nodes(ctx, (uint32_t)nb, in.data(), &prm, res.data()));
  std::vector<float> score(nb);
  out.resize(nb);
  for (size_t b = 0; b < nb; ++b) {
    project_alignment(out[b], *a1[b], *a2[b], jobs[b].z);
    score[b] = res[b].score;
    if (verbose >= 1) std::cerr << "Step: " << res[b].iterations << ", Violated: " << res[b].violated << std::endl;  // :1292
  }
  return score;
}

// --fold-aux reader, reference src/fold.cpp:230-259 ("> x" then "i j:p j:p ...", all 1-based)
void load_fold_aux(const std::string& file, const std::vector<Fasta>& fa, std::vector<BP>& bp) {
  std::ifstream is(file.c_str());
  if (!is.is_open()) throw strerror(errno);
  bp.assign(fa.size(), BP());
  for (size_t x = 0; x < fa.size(); ++x) bp[x].resize(fa[x].size());
  std::string s, t;
  uint x = 0, i, j;
  float p;
  while (std::getline(is, s)) {
    std::istringstream ss(s);
    if (!s.empty() && s[0] == '>') {
      ss >> t >> x;
      if (x < 1 || x > bp.size()) throw "fold-aux: sequence index out of range";
    } else {
      if (!(ss >> i) || x == 0) continue;
      if (i - 1 >= bp[x - 1].size()) bp[x - 1].resize(i);
      while (ss >> t)
        if (sscanf(t.c_str(), "%u:%f", &j, &p) == 2) bp[x - 1][i - 1].push_back(std::make_pair(j - 1, p));
    }
  }
  for (size_t k = 0; k < fa.size(); ++k)
    if (bp[k].size() != fa[k].size()) throw "fold-aux: row count does not match the sequence length";
}

void upload_bp(dafs_hip_ctx* ctx, const std::vector<BP>& bp) {
  std::vector<uint32_t> rowptr, col;
  std::vector<float> val;
  for (const BP& b : bp) {
    uint32_t n = 0;
    for (const SV& row : b) {
      rowptr.push_back(n);
      for (const auto& e : row) { col.push_back(e.first); val.push_back(e.second); ++n; }
    }
    rowptr.push_back(n);
  }
  check(dafs_hip_set_bp(ctx, rowptr.data(), col.data(), val.data()));
}

// writers in the formats the reference's readers accept (its own writers are compiled out,
// reference src/dafs.cpp:221-256); 1-based like load_bp / load_mp expect
void save_fold_aux(dafs_hip_ctx* ctx, const std::string& file, const std::vector<Fasta>& fa) {
  uint64_t nnz = 0, nrp = 0;
  check(dafs_hip_bp_result_size(ctx, 0, &nnz, &nrp));
  std::vector<uint32_t> rowptr(nrp), col(nnz);
  std::vector<float> val(nnz);
  check(dafs_hip_bp_fetch(ctx, 0, rowptr.data(), col.data(), val.data()));
  std::ofstream os(file.c_str());
  os.precision(9);
  size_t r = 0, e = 0;
  for (size_t x = 0; x < fa.size(); ++x) {
    os << "> " << x + 1 << std::endl;
    for (uint32_t i = 0; i < fa[x].size(); ++i) {
      os << i + 1;
      for (uint32_t k = rowptr[r + i]; k < rowptr[r + i + 1]; ++k) os << " " << col[e + k] + 1 << ":" << val[e + k];
      os << std::endl;
    }
    e += rowptr[r + fa[x].size()];
    r += fa[x].size() + 1;
  }
}
// --align-aux reader, reference src/align.cpp:204-246 ("> x y" then "i k:p k:p ...", all 1-based); the rows go to
// the device through dafs_hip_set_mp, which also lays out the transposes and computes the similarity scores
void load_align_aux(dafs_hip_ctx* ctx, const std::string& file, const std::vector<Fasta>& fa) {
  std::ifstream is(file.c_str());
  if (!is.is_open()) throw strerror(errno);
  const uint N = (uint)fa.size();
  std::vector<std::vector<std::vector<std::vector<std::pair<uint, float> > > > > mp(N);  // [x][y][i] -> (k, p)
  for (uint x = 0; x < N; ++x) {
    mp[x].resize(N);
    for (uint y = x + 1; y < N; ++y) mp[x][y].resize(fa[x].size());
  }
  std::string s, t;
  uint x = 0, y = 0;
  while (std::getline(is, s)) {
    if (s.empty()) continue;
    std::istringstream ss(s);
    if (s[0] == '>') {
      ss >> t >> x >> y;
      if (!(x < y && x >= 1 && y <= N)) throw "--align-aux: bad pair header";
    } else {
      uint i = 0, k = 0;
      float pr = 0;
      ss >> i;
      if (x == 0 || i < 1 || i > fa[x - 1].size()) throw "--align-aux: bad row index";
      while (ss >> t)
        if (sscanf(t.c_str(), "%u:%f", &k, &pr) == 2) {
          if (k < 1 || k > fa[y - 1].size()) throw "--align-aux: bad column index";
          mp[x - 1][y - 1][i - 1].push_back(std::make_pair(k - 1, pr));
        }
    }
  }
  std::vector<uint32_t> nnz, rowptr, col;
  std::vector<float> val;
  for (uint a = 0; a < N; ++a)
    for (uint b = a + 1; b < N; ++b) {
      uint32_t run = 0;
      rowptr.push_back(0);
      for (const auto& row : mp[a][b]) {
        for (const auto& e : row) { col.push_back(e.first); val.push_back(e.second); }
        run += (uint32_t)row.size();
        rowptr.push_back(run);
      }
      nnz.push_back(run);
    }
  if (col.empty()) { col.push_back(0); val.push_back(0.0f); }
  check(dafs_hip_set_mp(ctx, nnz.data(), rowptr.data(), col.data(), val.data()));
}

void save_align_aux(dafs_hip_ctx* ctx, const std::string& file, const std::vector<Fasta>& fa) {
  uint64_t np = 0, nnz = 0, nrp = 0;
  check(dafs_hip_mp_result_size(ctx, 0, &np, &nnz, &nrp));
  std::vector<uint32_t> px(np), py(np), cnt(np), rowptr(nrp), col(2 * nnz);
  std::vector<float> val(2 * nnz);
  check(dafs_hip_mp_fetch(ctx, 0, px.data(), py.data(), cnt.data(), rowptr.data(), col.data(), val.data()));
  std::ofstream os(file.c_str());
  os.precision(9);
  size_t r = 0, e = 0;
  for (uint64_t p = 0; p < np; ++p) {
    const uint32_t L1 = fa[px[p]].size(), L2 = fa[py[p]].size();
    os << "> " << px[p] + 1 << " " << py[p] + 1 << std::endl;
    for (uint32_t i = 0; i < L1; ++i) {
      os << i + 1;
      for (uint32_t k = rowptr[r + i]; k < rowptr[r + i + 1]; ++k) os << " " << col[e + k] + 1 << ":" << val[e + k];
      os << std::endl;
    }
    r += (size_t)L1 + 1 + L2 + 1;
    e += 2 * (size_t)cnt[p];
  }
}

// ---------------------------------------------------------------------------------------------
// --devices: the ranks of phase 1, one process per listed GPU.  The processes are forked before the first GPU call of
// the program (a process that has initialised the GPU must neither fork nor exec); rank 0 is the original process and
// the only one that goes on after phase 1.  What the ranks share is one anonymous mapping made before the fork.
struct RankShared {
  pthread_barrier_t barrier;
  ncclUniqueId nccl_id;  // written by rank 0, read by the others after a barrier
};

volatile sig_atomic_t g_child_count = 0;
pid_t g_child_pid[64];
volatile sig_atomic_t g_child_gone[64];

void on_sigchld(int) {  // a rank that fails takes the run down instead of leaving the others in a collective
  for (int k = 0; k < g_child_count; ++k) {
    if (g_child_gone[k]) continue;
    int st = 0;
    if (waitpid(g_child_pid[k], &st, WNOHANG) != g_child_pid[k]) continue;
    g_child_gone[k] = 1;
    if (!(WIFEXITED(st) && WEXITSTATUS(st) == 0)) {
      static const char msg[] = "dafs: a rank of --devices failed\n";
      if (write(2, msg, sizeof msg - 1) < 0) {}
      _exit(EXIT_FAILURE);  // the remaining ranks die with their parent (PR_SET_PDEATHSIG)
    }
  }
}

struct Ranks {
  uint32_t rank = 0, world = 1;
  RankShared* shared = nullptr;
  uint8_t* stage = nullptr;  // host staging area (only when a device is listed more than once)
  size_t stage_bytes = 0;
  void* lib = nullptr;
  ncclComm_t comm = nullptr;
  decltype(&ncclGetUniqueId) get_id = nullptr;
  decltype(&ncclCommInitRank) comm_init = nullptr;
  decltype(&ncclAllGather) all_gather = nullptr;
  decltype(&ncclCommDestroy) comm_destroy = nullptr;
  decltype(&ncclGetErrorString) err_string = nullptr;

  void fork_ranks(const std::vector<int>& devices) {
    world = (uint32_t)devices.size();
    if (world > 64) throw std::string("--devices: at most 64 devices");
    bool repeated = false;
    for (size_t a = 0; a < devices.size(); ++a)
      for (size_t b = 0; b < a; ++b) repeated |= devices[a] == devices[b];
    // RCCL refuses two ranks on one device; such a list (a rehearsal of the multi-process path on a single GPU) exchanges
    // its shards through a host staging area instead.  Untouched pages of the mapping cost nothing.
    if (repeated) stage_bytes = (size_t)sysconf(_SC_PHYS_PAGES) * (size_t)sysconf(_SC_PAGE_SIZE) / 4;
    const size_t head = 4096;
    static_assert(sizeof(RankShared) <= 4096, "the header page");
    void* m = mmap(nullptr, head + stage_bytes, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (m == MAP_FAILED) throw std::string("--devices: cannot map the shared area");
    shared = (RankShared*)m;
    stage = (uint8_t*)m + head;
    pthread_barrierattr_t at;
    pthread_barrierattr_init(&at);
    pthread_barrierattr_setpshared(&at, PTHREAD_PROCESS_SHARED);
    pthread_barrier_init(&shared->barrier, &at, world);
    std::cout.flush();
    std::cerr.flush();
    const pid_t parent = getpid();
    for (uint32_t r = 1; r < world; ++r) {
      const pid_t pid = fork();
      if (pid < 0) throw std::string("--devices: fork failed");
      if (pid == 0) {
        prctl(PR_SET_PDEATHSIG, SIGKILL);
        if (getppid() != parent) _exit(EXIT_FAILURE);  // the parent died before the prctl
        rank = r;
        g_child_count = 0;
        return;
      }
      g_child_pid[r - 1] = pid;
      g_child_gone[r - 1] = 0;
      g_child_count = (sig_atomic_t)r;
    }
    struct sigaction sa;
    memset(&sa, 0, sizeof sa);
    sa.sa_handler = on_sigchld;
    sa.sa_flags = SA_RESTART | SA_NOCLDSTOP;
    sigaction(SIGCHLD, &sa, nullptr);
    on_sigchld(0);  // a rank that died before the handler was in place
  }

  void barrier() { if (world > 1) pthread_barrier_wait(&shared->barrier); }

  void connect() {  // after dafs_hip_create: the communicator belongs to the context's device
    if (stage_bytes) return;
    lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!lib) throw std::string("--devices: cannot load librccl.so.1: ") + dlerror();
    get_id = (decltype(get_id))dlsym(lib, "ncclGetUniqueId");
    comm_init = (decltype(comm_init))dlsym(lib, "ncclCommInitRank");
    all_gather = (decltype(all_gather))dlsym(lib, "ncclAllGather");
    comm_destroy = (decltype(comm_destroy))dlsym(lib, "ncclCommDestroy");
    err_string = (decltype(err_string))dlsym(lib, "ncclGetErrorString");
    if (!get_id || !comm_init || !all_gather || !comm_destroy || !err_string) throw std::string("--devices: librccl.so.1 lacks an entry point");
    // RCCL announces itself on stdout when the communicator is made (version, host, library path): this program's
    // stdout is its result, so that goes to stderr
    std::cout.flush();
    fflush(stdout);
    const int keep = dup(1);
    if (keep >= 0) dup2(2, 1);
    ncclResult_t r0 = ncclSuccess, r1 = ncclSuccess;
    if (rank == 0) r0 = get_id(&shared->nccl_id);
    barrier();
    if (r0 == ncclSuccess) r1 = comm_init(&comm, (int)world, shared->nccl_id, (int)rank);
    fflush(stdout);
    if (keep >= 0) { dup2(keep, 1); close(keep); }
    nccl(r0);
    nccl(r1);
  }
  void disconnect() {
    if (comm) { comm_destroy(comm); comm = nullptr; }
  }
  void nccl(ncclResult_t r) {
    if (r != ncclSuccess) throw std::string("RCCL: ") + err_string(r);
  }
  void wait_ranks() {  // rank 0, at the end: every rank has left without an error
    sigset_t block, old;
    sigemptyset(&block);
    sigaddset(&block, SIGCHLD);
    sigprocmask(SIG_BLOCK, &block, &old);
    bool bad = false;
    for (int k = 0; k < g_child_count; ++k) {
      if (g_child_gone[k]) continue;
      int st = 0;
      if (waitpid(g_child_pid[k], &st, 0) == g_child_pid[k]) bad |= !(WIFEXITED(st) && WEXITSTATUS(st) == 0);
      g_child_gone[k] = 1;
    }
    sigprocmask(SIG_SETMASK, &old, nullptr);
    if (bad) throw std::string("a rank of --devices failed");
  }
};

// dafs_allgather_fn: ncclAllGather on the stream the library names; between two ranks of one device, host staging
int rank_allgather(void* user, const void* send, void* recv, size_t bytes, void* stream) {
  Ranks* rk = (Ranks*)user;
  if (rk->comm) {
    const ncclResult_t r = rk->all_gather(send, recv, bytes, ncclChar, rk->comm, (hipStream_t)stream);
    if (r != ncclSuccess) std::cerr << "RCCL: " << rk->err_string(r) << std::endl;
    return r == ncclSuccess ? 0 : 1;
  }
  if (bytes * rk->world > rk->stage_bytes) {
    std::cerr << "--devices: the host staging area is too small for " << bytes << " bytes per rank" << std::endl;
    return 1;
  }
  if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return 1;
  if (hipMemcpy(rk->stage + (size_t)rk->rank * bytes, send, bytes, hipMemcpyDeviceToHost) != hipSuccess) return 1;
  rk->barrier();
  if (hipMemcpy(recv, rk->stage, (size_t)rk->world * bytes, hipMemcpyHostToDevice) != hipSuccess) return 1;
  rk->barrier();  // nobody overwrites the area before everybody has read it
  return 0;
}

int run(const Options& o, Ranks& rk) {
  // ---- option checks mirroring parse_options (:1683-1763)
  int align_model;
  if (o.align_model == "ProbCons") align_model = DAFS_ALIGN_PROBCONS;
  else if (o.align_model == "CONTRAlign") align_model = DAFS_ALIGN_CONTRALIGN;
  else throw "Unknown alignment model: " + o.align_model;
  if (o.fold_aux.empty()) {
    if (o.fold_model == "Boltzmann" || o.fold_model == "Vienna")
      throw "Folding model " + o.fold_model + " needs ViennaRNA, which this build does not contain; use -s CONTRAfold or --fold-aux";
    if (o.fold_model != "CONTRAfold") throw "Unknown folding model: " + o.fold_model;
  }
  if (o.fold_decoder != "Nussinov" || o.ipknot) throw "Folding decoder IPknot needs an ILP solver, which this build does not contain";
  if (o.max_iter <= 0) throw "-m 0 (exact ILP) needs an ILP solver, which this build does not contain";
  if ((o.bp_update || o.bp_update1) && !o.fold_aux.empty()) throw "--bp-update / --bp-update1 need a folding model (-s CONTRAfold), not --fold-aux";
  if (!o.devices.empty() && (!o.fold_aux.empty() || !o.align_aux.empty() || o.fourway != 0.0f))
    throw std::string("--devices shards the posterior models and the consistency transform; --fold-aux, --align-aux and -f run on one device (--device)");
  if (o.verbose >= 1) {
    if (!o.no_alifold) std::cerr << "note: RNAalifold is not available in this build; running as with --no-alifold" << std::endl;
    if (!o.fold_model_given && o.fold_aux.empty()) std::cerr << "note: default folding model is CONTRAfold in this build" << std::endl;
  }

  std::vector<Fasta> fa;
  Fasta::load(fa, o.input.c_str());
  const uint N = (uint)fa.size();
  if (N == 0) throw "no sequences in the input";

  dafs_hip_ctx* ctx = nullptr;
  check(dafs_hip_create(o.devices.empty() ? o.device : o.devices[rk.rank], &ctx));
  struct Guard { dafs_hip_ctx* c; ~Guard() { dafs_hip_destroy(c); } } guard{ctx};

  std::vector<const char*> seqs(N);
  std::vector<uint32_t> lens(N);
  for (uint i = 0; i < N; ++i) { seqs[i] = fa[i].seq().c_str(); lens[i] = fa[i].size(); }
  check(dafs_hip_set_sequences(ctx, N, seqs.data(), lens.data()));

  // base-pairing probabilities (:1787).  The device folding is only started here: it keeps one workgroup per
  // sequence busy, and the alignment posteriors and the matching-probability transform run beside it.
  const bool sharded = !o.devices.empty() && N > 1;
  if (!sharded && rk.rank != 0) return 0;  // a single sequence: nothing to share
  bool folding = false;
  if (sharded) {
  } else if (!o.fold_aux.empty()) {
    std::vector<BP> bp;
    load_fold_aux(o.fold_aux, fa, bp);
    upload_bp(ctx, bp);
  } else {
    check(dafs_hip_fold_posteriors_begin(ctx, DAFS_FOLD_CONTRAFOLD, kCutoff));
    folding = true;
  }
  bool fold_saved = false;
  auto finish_folding = [&]() {
    if (folding) { check(dafs_hip_fold_posteriors_end(ctx)); folding = false; }
    if (!o.save_fold_aux.empty() && !fold_saved) { save_fold_aux(ctx, o.save_fold_aux, fa); fold_saved = true; }
  };

  std::vector<node_t> tree(1, std::make_pair(0.0f, std::make_pair(-1u, -1u)));
  if (N == 1) finish_folding();
  if (sharded) {
    // phase 1 (:1787-1827) on rk.world ranks: every rank ends with the complete stores, rank 0 goes on alone
    rk.connect();
    check(dafs_hip_phase1_sharded(ctx, rk.rank, rk.world, align_model, o.align_th, o.align_pct, o.fold_pct, DAFS_FOLD_CONTRAFOLD, kCutoff, rank_allgather, &rk));
    rk.disconnect();
    if (rk.rank != 0) return 0;
    if (!o.save_fold_aux.empty()) save_fold_aux(ctx, o.save_fold_aux, fa);
    if (!o.save_align_aux.empty()) save_align_aux(ctx, o.save_align_aux, fa);
    std::vector<float> sim((size_t)N * N);
    check(dafs_hip_get_sim(ctx, sim.data()));
    tree = build_tree(sim, N);
  } else if (N > 1) {
    // matching probabilities, transposes, similarities (:1796-1819), PCTs (:1822-1827), tree (:1830)
    if (!o.align_aux.empty()) load_align_aux(ctx, o.align_aux, fa);
    else check(dafs_hip_align_posteriors(ctx, align_model, o.align_th, 0, 0));
    if (!o.save_align_aux.empty()) save_align_aux(ctx, o.save_align_aux, fa);
    if (o.fourway != 0.0f) {  // relax_fourway_consistency (:1808-1809): needs the base-pairing rows, replaces mp_ before sim_
      finish_folding();
      check(dafs_hip_fourway_consistency(ctx, o.fourway));
    }
    std::vector<float> sim((size_t)N * N);
    check(dafs_hip_get_sim(ctx, sim.data()));
    check(dafs_hip_consistency_match(ctx, o.align_pct));
    finish_folding();
    check(dafs_hip_consistency_bp(ctx, o.fold_pct));
    tree = build_tree(sim, N);
  }
  print_tree(std::cout, tree, fa, (int)tree.size() - 1);
  std::cout << std::endl;

  // progressive alignment (:1838): every node whose children are ready is solved in the same batch
  dafs_dd_params prm;
  dafs_hip_dd_default_params(&prm);
  prm.w = o.w; prm.eta0 = o.eta; prm.th_a = o.align_th; prm.th_s = *std::min_element(o.fold_th.begin(), o.fold_th.end());
  prm.t_max = (uint32_t)o.max_iter;
  // The progressive loop below uses the alignment of a node and nothing else (dafs.cpp:896-912), except that the
  // score of the root seeds the iterative refinement and -v prints every node's iteration count (:1292): without
  // either, nodes with no consensus base pair may leave out their folding DPs
  // (dafs_dd_params::skip_uncoupled_folds).  The refinement itself compares scores.
  dafs_dd_params prm_prog = prm;
  prm_prog.skip_uncoupled_folds = (o.refinement == 0 && o.verbose == 0) ? 1 : 0;
  std::vector<ALN> aln(tree.size());
  std::vector<bool> done(tree.size(), false);
  for (uint i = 0; i < N; ++i) {
    aln[i].push_back(std::make_pair(i, std::vector<bool>(fa[i].size(), true)));
    done[i] = true;
  }
  float s = 0.0f;
  {
    // The nodes stay resident on the device (dafs_hip_nodes_*).  A round is one call (dafs_hip_nodes_round): the open
    // nodes advance while the nodes whose children have just finished are set up and started beside them, and all of
    // them stop together after kRoundUs microseconds, so a node that needs the full iteration budget does not hold
    // back its level and the set-up of new nodes does not stand between two launches.
    const uint32_t kRoundUs = getenv("DAFS_ROUND_US") ? (uint32_t)atoi(getenv("DAFS_ROUND_US")) : 2500u;
    struct Open { uint node; uint32_t handle; NodeJob job; };
    std::vector<Open> open;
    size_t remaining = tree.size() - N;
    std::vector<bool> opened(tree.size(), false);
    while (remaining) {
      std::vector<uint> ready;
      for (uint i = N; i < tree.size(); ++i)
        if (!done[i] && !opened[i] && done[tree[i].second.first] && done[tree[i].second.second]) ready.push_back(i);
      const size_t n_old = open.size();
      std::vector<dafs_node_input> in(ready.size() ? ready.size() : 1);
      for (size_t b = 0; b < ready.size(); ++b) {
        open.push_back(Open{ready[b], 0, NodeJob()});
        opened[ready[b]] = true;
      }
      for (size_t b = 0; b < ready.size(); ++b) {
        const ALN &a1 = aln[tree[ready[b]].second.first], &a2 = aln[tree[ready[b]].second.second];
        NodeJob& j = open[n_old + b].job;
        flatten(a1, j.s1, j.m1);
        flatten(a2, j.s2, j.m2);
        in[b].n1 = (uint32_t)a1.size(); in[b].n2 = (uint32_t)a2.size();
        in[b].len1 = (uint32_t)a1[0].second.size(); in[b].len2 = (uint32_t)a2[0].second.size();
        in[b].seq1 = j.s1.data(); in[b].seq2 = j.s2.data(); in[b].mask1 = j.m1.data(); in[b].mask2 = j.m2.data();
        j.x.resize(in[b].len1); j.y.resize(in[b].len2); j.z.resize(in[b].len1);
        if (o.bp_update && ready[b] == tree.size() - 1) {
          // the top call of the recursion re-estimates both base-pairing matrices under the structure decoded from
          // their averages (align_alignments(ss, ...), :919-934)
          j.px.resize((size_t)in[b].len1 * in[b].len1);
          j.py.resize((size_t)in[b].len2 * in[b].len2);
          updated_bp(ctx, in[b].n1, in[b].len1, j.s1, j.m1, prm.th_s, j.px);
          updated_bp(ctx, in[b].n2, in[b].len2, j.s2, j.m2, prm.th_s, j.py);
          in[b].p_x = j.px.data(); in[b].p_y = j.py.data();
        }
      }
      std::vector<uint32_t> old_handles(n_old ? n_old : 1), new_handles(ready.size() ? ready.size() : 1);
      std::vector<uint8_t> fin(open.size() ? open.size() : 1, 0);
      for (size_t k = 0; k < n_old; ++k) old_handles[k] = open[k].handle;
      check(dafs_hip_nodes_round(ctx, (uint32_t)ready.size(), in.data(), new_handles.data(), (uint32_t)n_old, old_handles.data(), &prm_prog, 0,
                                 kRoundUs, fin.data(), fin.data() + n_old));
      for (size_t b = 0; b < ready.size(); ++b) open[n_old + b].handle = new_handles[b];
      std::vector<Open> still;
      for (size_t k = 0; k < open.size(); ++k) {
        if (!fin[k]) { still.push_back(std::move(open[k])); continue; }
        Open& o1 = open[k];
        dafs_node_output r;
        r.x = o1.job.x.data(); r.y = o1.job.y.data(); r.z = o1.job.z.data();
        check(dafs_hip_nodes_result(ctx, o1.handle, &r));
        const uint l = tree[o1.node].second.first, rr = tree[o1.node].second.second;
        ALN merged;
        project_alignment(merged, aln[l], aln[rr], o1.job.z);
        aln[o1.node].swap(merged);
        done[o1.node] = true;
        ALN().swap(aln[l]);
        ALN().swap(aln[rr]);
        if (o.verbose >= 1) std::cerr << "Step: " << r.iterations << ", Violated: " << r.violated << std::endl;  // :1292
        if (o1.node == tree.size() - 1) s = r.score;
        --remaining;
      }
      open.swap(still);
    }
    check(dafs_hip_nodes_close(ctx));
  }
  ALN& root = aln[tree.size() - 1];

  // iterative refinement (:1841-1855, refine :1539-1576; rand() is unseeded there too)
  for (int it = 0; it < o.refinement && root.size() > 1; ++it) {
    VU group[2];
    do {
      group[0].clear();
      group[1].clear();
      for (uint i = 0; i != root.size(); ++i) group[rand() % 2].push_back(i);
    } while (group[0].empty() || group[1].empty());
    ALN part[2];
    for (uint g = 0; g != 2; ++g) {
      const uint n = (uint)group[g].size(), L = (uint)root[group[g][0]].second.size();
      part[g].resize(n);
      for (uint j = 0; j != n; ++j) part[g][j].first = root[group[g][j]].first;
      for (uint k = 0; k != L; ++k) {
        bool gap = true;
        for (uint j = 0; j != n; ++j) gap &= !root[group[g][j]].second[k];
        if (!gap)
          for (uint j = 0; j != n; ++j) part[g][j].second.push_back(root[group[g][j]].second[k]);
      }
    }
    std::vector<ALN> merged;
    const std::vector<float> sc = solve_batch(ctx, prm, {&part[0]}, {&part[1]}, merged, o.verbose, o.bp_update);
    if (sc[0] > s) { s = sc[0]; root.swap(merged[0]); }
  }

  // common secondary structure of the final alignment (:1857-1871; no RNAalifold term here)
  std::string str;
  {
    std::vector<uint32_t> rs;
    std::vector<uint8_t> rm;
    flatten(root, rs, rm);
    const uint32_t L = (uint32_t)root[0].second.size();
    VU ss(L);
    check(dafs_hip_consensus_structure(ctx, (uint32_t)root.size(), L, rs.data(), rm.data(), o.fold_th1[0], ss.data(), nullptr, nullptr));
    if (o.bp_update1) {  // :1863-1869: re-estimate under the decoded structure, decode again (SparseNussinov::decode(p, ss, str))
      std::vector<float> p((size_t)L * L);
      check(dafs_hip_update_basepairing(ctx, (uint32_t)root.size(), L, rs.data(), rm.data(), ss.data(), p.data()));
      check(dafs_hip_nussinov_decode(ctx, o.fold_th1[0], 0.0f, L, p.data(), nullptr, ss.data(), nullptr));
    }
    std::vector<char> buf(L + 1);
    dafs_hip_make_brackets(L, ss.data(), buf.data());
    str.assign(buf.data());
  }

  // output (:1876-1879, :1584-1601)
  std::sort(root.begin(), root.end());
  std::cout << ">SS_cons" << std::endl << str << std::endl;
  for (const auto& row : root) {
    const std::string& sq = fa[row.first].seq();
    std::cout << "> " << fa[row.first].name() << std::endl;
    for (uint j = 0, k = 0; j != row.second.size(); ++j) std::cout << (row.second[j] ? sq[k++] : '-');
    std::cout << std::endl;
  }
  return 0;
}

}  // namespace

int main(int argc, char* argv[]) {
  try {
    const Options o = parse(argc, argv);
    Ranks rk;
    if (!o.devices.empty()) rk.fork_ranks(o.devices);  // before the first GPU call
    const int rc = run(o, rk);
    if (rk.rank == 0) rk.wait_ranks();
    return rc;
  } catch (const char* str) {
    std::cerr << str << std::endl;
  } catch (const std::string& str) {
    std::cerr << str << std::endl;
  } catch (const std::system_error& e) {
    std::cerr << e.what() << std::endl;
  } catch (const std::exception& e) {
    std::cerr << e.what() << std::endl;
  }
  return EXIT_FAILURE;
}
