// dafs_amd/csrc/host/plugin_selftest.cpp -- drives every method of the plugin classes of plugins.h
// exactly as the reference's DAFS class drives its plugins (reference src/dafs.cpp:1787,1796,1064,
// 1091-1093,1867-1870) and prints the results with exact float bits, for tests/test_cli_gpu.py to
// compare with the oracle.  Usage: plugin_selftest FASTA
#include <cstdio>
#include <cstring>
#include <iostream>
#include <memory>

#include "plugins.h"

static unsigned bits(float f) { unsigned u; std::memcpy(&u, &f, 4); return u; }

static void dump_rows(const char* tag, unsigned a, unsigned b, const std::vector<SV>& rows) {
  for (size_t i = 0; i < rows.size(); ++i)
    for (const auto& e : rows[i]) std::printf("%s %u %u %zu %u %08x\n", tag, a, b, i, e.first, bits(e.second));
}

int main(int argc, char** argv) {
  try {
    if (argc < 2) { std::fprintf(stderr, "usage: plugin_selftest FASTA\n"); return 2; }
    std::vector<Fasta> fa;
    Fasta::load(fa, argv[1]);
    auto ctx = std::make_shared<HipContext>(0);
    const float CUTOFF = 0.01f;
    // Fold::Model: batch, single, constrained
    HipCONTRAfold fold(ctx, CUTOFF);
    std::vector<BP> bp;
    fold.calculate(fa, bp);
    for (size_t x = 0; x < bp.size(); ++x) dump_rows("BP", (unsigned)x, 0, bp[x]);
    BP one;
    fold.calculate(fa[0].seq(), one);
    dump_rows("BP1", 0, 0, one);
    std::string cons(fa[0].size(), '?');
    cons[0] = '('; cons[cons.size() - 1] = ')'; cons[3] = '.';
    fold.calculate(fa[0].seq(), cons, one);
    dump_rows("BPC", 0, 0, one);
    // Align::Model: batch + single, both models
    for (int model = 0; model < 2; ++model) {
      HipAlignModel am(ctx, model, 0.01f);
      std::vector<std::vector<MP> > mp;
      am.calculate(fa, mp);
      for (size_t i = 0; i < fa.size(); ++i)
        for (size_t j = i; j < fa.size(); ++j) dump_rows(model ? "MPC" : "MPP", (unsigned)i, (unsigned)j, mp[i][j]);
      MP m01;
      am.calculate(fa[0].seq(), fa[1].seq(), m01);
      dump_rows(model ? "MPC1" : "MPP1", 0, 1, m01);
      if (model == 0) {
        // Align::Decoder on the dense mp[0][1]
        const size_t L1 = fa[0].size(), L2 = fa[1].size();
        VVF p(L1, VF(L2, 0.0f)), q(L1, VF(L2, 0.0f));
        for (size_t i = 0; i < L1; ++i)
          for (const auto& e : mp[0][1][i]) { p[i][e.first] = e.second; q[i][e.first] = 0.125f * (float)((i + e.first) % 3); }
        HipSparseNeedlemanWunsch nw(ctx, 0.01f);
        nw.initialize(p);
        VU al;
        float s = nw.decode(p, q, al);
        std::printf("NWQ %08x", bits(s));
        for (uint v : al) std::printf(" %d", (int)v);
        std::printf("\n");
        s = nw.decode(p, al);
        std::printf("NW %08x", bits(s));
        for (uint v : al) std::printf(" %d", (int)v);
        std::printf("\n");
      }
    }
    // Fold::Decoder on the dense bp[0]
    {
      const size_t L = fa[0].size();
      VVF p(L, VF(L, 0.0f)), q(L, VF(L, 0.0f));
      for (size_t i = 0; i < L; ++i)
        for (const auto& e : bp[0][i]) { p[i][e.first] = e.second; q[i][e.first] = 0.25f * (float)((i * 7 + e.first) % 4) - 0.25f; }
      HipSparseNussinov nu(ctx, 0.2f);
      VU ss;
      float s = nu.decode(4.0f, p, q, ss);
      std::printf("NUQ %08x", bits(s));
      for (uint v : ss) std::printf(" %d", (int)v);
      std::printf("\n");
      std::string str;
      s = nu.decode(p, ss, str);
      std::printf("NU %08x %s\n", bits(s), str.c_str());
    }
    return 0;
  } catch (const char* msg) {
    std::fprintf(stderr, "%s\n", msg);
  } catch (const std::exception& e) {
    std::fprintf(stderr, "%s\n", e.what());
  }
  return 1;
}
