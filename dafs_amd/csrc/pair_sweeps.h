// dafs_amd/csrc/pair_sweeps.h -- the model-independent tail of the pair-posterior kernels.
//
// Both alignment models (ProbCons pairhmm3.hip, CONTRAlign pairhmm5.hip) leave one float per DP
// cell in plane 0 of the wave's slab, indexed [(step*W + c)*64 + lane] in the skewed layout
// described in pairhmm3.hip.  pair_finish turns that plane into the sparse outputs:
//   sweep 3: p = post(slab value); threshold (wrapper >= th, adapter > th: reference
//            src/align.cpp:69-78); similarity-score DP (src/dafs.cpp:713-764); entry counts per
//            row (carried lane to lane with the row) and per column (registers)
//   sweep 4: scatter into the CSR of mp[x][y] and of mp[y][x] (src/dafs.cpp:155-167)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/dafs_hip.h"
#include "hip_util.h"

namespace dafs {

// Neighbour exchange inside a group of G lanes: value of lane t-1 (shift_up1) or t+1 (shift_down1),
// `fill` at the group boundary.  For G = 16 a group is one DPP row, so the move is a single
// row_shr:1 / row_shl:1 VALU instruction with the boundary fill for free.
template <int G>
__device__ __forceinline__ float shift_up1(float v, float fill, int t) {
  if constexpr (G == 16) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(fill), __float_as_int(v), 0x111, 0xF, 0xF, false));
  } else {
    const float r = __shfl_up(v, 1, G);
    return t == 0 ? fill : r;
  }
}
template <int G>
__device__ __forceinline__ int shift_up1(int v, int fill, int t) {
  if constexpr (G == 16) {
    return __builtin_amdgcn_update_dpp(fill, v, 0x111, 0xF, 0xF, false);
  } else {
    const int r = __shfl_up(v, 1, G);
    return t == 0 ? fill : r;
  }
}
template <int G>
__device__ __forceinline__ float shift_down1(float v, float fill, int t) {
  if constexpr (G == 16) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(fill), __float_as_int(v), 0x101, 0xF, 0xF, false));
  } else {
    const float r = __shfl_down(v, 1, G);
    return t == G - 1 ? fill : r;
  }
}

template <int G, int W, class Args, class Post>
__device__ __forceinline__ void pair_finish(const Args& a, float* __restrict__ slab, uint32_t* __restrict__ s_rowptr, int lane, int t, int g,
                                            int L1, int L2, int nsteps, int tlast, bool act, uint32_t task, float th, Post post) {
  // ------------------------------------------------------------------ sweep 3: posterior + sim + counts
  int colcnt[W];
  float simv = 0.0f;
  uint32_t nnz = 0;
  {
    float pdp[W];
    int ptr[W];
#pragma unroll
    for (int c = 0; c < W; ++c) { pdp[c] = 0.0f; ptr[c] = 0; colcnt[c] = 0; }
    float lastdp = 0.0f, dgdp = 0.0f;
    int lasttr = 0, dgtr = 0, lastcnt = 0;
    uint32_t rowacc = 0;
    if (t == G - 1) s_rowptr[0] = 0;
    for (int s = 0; s < nsteps; ++s) {
      const int i = s - t;
      const bool rowv = (i >= 0) && (i <= L1);
      const float rdp = shift_up1<G>(lastdp, 0.0f, t);
      const int rtr = shift_up1<G>(lasttr, 0, t), rcnt = shift_up1<G>(lastcnt, 0, t);
      float ddp = dgdp, ldp = rdp;
      int dtr = dgtr, ltr = rtr, run = rcnt;
      float sv[W];
#pragma unroll
      for (int c = 0; c < W; ++c) {
        sv[c] = slab[(size_t)(s * W + c) * 64 + lane];  // private slot: unguarded (cells outside the grid are ignored below)
      }
#pragma unroll
      for (int c = 0; c < W; ++c) {
        const int j = t * W + c;
        const bool v = rowv && (j <= L2);
        const bool inner = v && i >= 1 && j >= 1;
        // the model's posterior; wrapper (>= th keeps) then adapter (> th keeps): align.cpp:69-78
        const float p = post(sv[c]);
        const bool entry = inner && (p >= th) && (p > th);
        slab[(size_t)(s * W + c) * 64 + lane] = entry ? p : 0.0f;
        // calculate_similarity_score, dafs.cpp:720-760
        const float udp = pdp[c];
        const int utr = ptr[c];
        float dp;
        int tr;
        if (entry) {
          dp = ddp + p; tr = dtr + 1;
          if (dp < ldp) { dp = ldp; tr = ltr + 1; }
          if (dp < udp) { dp = udp; tr = utr + 1; }
        } else {
          dp = ldp; tr = ltr + 1;
          if (dp < udp) { dp = udp; tr = utr + 1; }
        }
        if (!inner) { dp = 0.0f; tr = 0; }
        ddp = udp; dtr = utr;
        pdp[c] = dp; ptr[c] = tr;
        ldp = dp; ltr = tr;
        run += entry ? 1 : 0;
        colcnt[c] += entry ? 1 : 0;
        if (v && i == L1 && j == L2) simv = dp / (float)tr;  // dafs.cpp:763
      }
      dgdp = rdp; dgtr = rtr;
      lastdp = pdp[W - 1]; lasttr = ptr[W - 1]; lastcnt = run;
      if (t == G - 1 && rowv && i >= 1) {  // row i is complete: its count has crossed the group
        rowacc += (uint32_t)run;
        s_rowptr[i] = rowacc;
      }
    }
    nnz = rowacc;
  }
  nnz = __shfl(nnz, g * G + (G - 1));
  simv = __shfl(simv, g * G + tlast);

  // column prefix sums (row pointers of the transposed matrix)
  int colbase[W];
  {
    int mine = 0;
#pragma unroll
    for (int c = 0; c < W; ++c) mine += colcnt[c];
    int incl = mine;
#pragma unroll
    for (int o = 1; o < G; o <<= 1) {
      const int up = __shfl_up(incl, o, G);
      if (t >= o) incl += up;
    }
    int run = incl - mine;
#pragma unroll
    for (int c = 0; c < W; ++c) { colbase[c] = run; run += colcnt[c]; }
  }

  // reserve 2*nnz entries in the pool
  unsigned long long off = 0;
  if (t == 0 && act) off = atomicAdd(a.pool_top, 2ull * nnz);
  off = __shfl(off, g * G);
  const bool ok = act && (off + 2ull * nnz <= a.pool_cap);
  if (act && !ok && t == 0) atomicExch(a.status, DAFS_HIP_EOVERFLOW);
  const uint64_t rp = act ? a.rp_off[task] : 0;
  if (act && t == 0) {
    a.pair_off[task] = off;
    a.pair_nnz[task] = nnz;
    a.sim[task] = simv;
  }
  wave_lds_fence();
  // row pointers out (coalesced copy from LDS), transposed row pointers from registers
  if (act) {
    for (int r = t; r <= L1; r += G) a.rowptr_pool[rp + r] = s_rowptr[r];
    if (t == 0) a.rowptr_pool[rp + L1 + 1] = 0;
#pragma unroll
    for (int c = 0; c < W; ++c) {
      const int j = t * W + c;
      if (j >= 1 && j <= L2) a.rowptr_pool[rp + L1 + 1 + j] = (uint32_t)(colbase[c] + colcnt[c]);
    }
  }

  // ------------------------------------------------------------------ sweep 4: emit CSR + transposed CSR
  if (__any(ok)) {
    int colrun[W];
#pragma unroll
    for (int c = 0; c < W; ++c) colrun[c] = 0;
    int lastcnt = 0;
    for (int s = 0; s < nsteps; ++s) {
      const int i = s - t;
      const bool rowv = (i >= 0) && (i <= L1);
      int run = shift_up1<G>(lastcnt, 0, t);
      const uint32_t rowbase = (rowv && i >= 1) ? s_rowptr[i - 1] : 0;
      float pv[W];
#pragma unroll
      for (int c = 0; c < W; ++c) {
        pv[c] = slab[(size_t)(s * W + c) * 64 + lane];  // sweep 3 left 0 in every non-entry slot
      }
#pragma unroll
      for (int c = 0; c < W; ++c) {
        const int j = t * W + c;
        const bool entry = pv[c] != 0.0f;
        if (entry && ok) {
          const unsigned long long pos = off + rowbase + (uint32_t)run;
          a.ent_col[pos] = (uint32_t)(j - 1);
          a.ent_val[pos] = pv[c];
          const unsigned long long tpos = off + nnz + (uint32_t)(colbase[c] + colrun[c]);
          a.ent_col[tpos] = (uint32_t)(i - 1);
          a.ent_val[tpos] = pv[c];
        }
        run += entry ? 1 : 0;
        colrun[c] += entry ? 1 : 0;
      }
      lastcnt = run;
    }
  }
  wave_lds_fence();
}

}  // namespace dafs
