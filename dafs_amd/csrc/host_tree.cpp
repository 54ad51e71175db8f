// dafs_amd/csrc/host_tree.cpp -- DAFS::build_tree (reference src/dafs.cpp:446-492) as a C entry point, so that
// every host (the C++ command line, the Python driver of tests and bench.py) builds the guide tree with the same
// code: greedy joins taken from a max-heap of (similarity, (i, j)), merged distance (d[ii][l] + d[ii][r]) * s / 2.
// Host logic only; nothing here touches the device.
#include <queue>
#include <utility>
#include <vector>

#include "../../include/dafs_hip.h"

extern "C" int dafs_host_build_tree(uint32_t n0, const float* sim, float* score, int32_t* left, int32_t* right) {
  if (!n0 || !sim || !score || !left || !right) return DAFS_HIP_EINVAL;
  typedef std::pair<float, std::pair<uint32_t, uint32_t> > node_t;
  uint32_t n = n0;
  const uint32_t T = 2 * n0 - 1;
  for (uint32_t i = 0; i < T; ++i) { score[i] = 0.0f; left[i] = -1; right[i] = -1; }
  std::vector<std::vector<float> > d(n, std::vector<float>(n, 0.0f));
  std::vector<uint32_t> idx(T, UINT32_MAX);
  for (uint32_t i = 0; i != n; ++i) idx[i] = i;
  std::priority_queue<node_t> pq;
  for (uint32_t i = 0; i + 1 < n; ++i)
    for (uint32_t j = i + 1; j != n; ++j) {
      d[i][j] = d[j][i] = sim[(size_t)i * n0 + j];
      pq.push(std::make_pair(sim[(size_t)i * n0 + j], std::make_pair(i, j)));
    }
  while (!pq.empty()) {
    const node_t t = pq.top();
    pq.pop();
    const uint32_t a = t.second.first, b = t.second.second;
    if (idx[a] == UINT32_MAX || idx[b] == UINT32_MAX) continue;
    const uint32_t l = idx[a], r = idx[b];
    idx[a] = idx[b] = UINT32_MAX;
    for (uint32_t i = 0; i != n; ++i)
      if (idx[i] != UINT32_MAX) {
        const uint32_t ii = idx[i];
        d[ii][l] = d[l][ii] = (d[ii][l] + d[ii][r]) * t.first / 2;
        pq.push(std::make_pair(d[ii][l], std::make_pair(i, n)));
      }
    score[n] = t.first;
    left[n] = (int32_t)a;
    right[n] = (int32_t)b;
    idx[n++] = l;
  }
  return DAFS_HIP_OK;
}
