// dafs_amd/csrc/pair_sweeps.h -- the model-independent parts of the pair-posterior kernels.
//
// Both alignment models (ProbCons pairhmm3.hip, CONTRAlign pairhmm5.hip) leave one float per DP
// cell in plane 0 of the wave's slab, indexed [(step*W + c)*64 + lane] in the skewed layout
// described in pairhmm3.hip.  pair_finish turns that plane into the sparse outputs:
//   sweep 3: p = post(slab value); threshold (wrapper >= th, adapter > th: reference
//            src/align.cpp:69-78); similarity-score DP (src/dafs.cpp:713-764); entry counts per
//            row (carried lane to lane with the row) and per column (registers)
//   sweep 4: scatter into the CSR of mp[x][y] and of mp[y][x] (src/dafs.cpp:155-167)
//
// Column ownership.  A kernel instance is compiled for W columns per lane, but a wavefront whose
// pairs are short enough runs with wr = W-1 (lane t owns columns t*wr .. t*wr+wr-1): the last cell
// of every step is then skipped by a wave-uniform branch.  Lengths inside one batch differ by a few
// per cent (the benchmark sets: +-7 %), and with W-1 = 5 instead of 6 columns that is 17 % of the cells.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "../../include/dafs_hip.h"
#include "hip_util.h"

namespace dafs {

// Neighbour exchange inside a group of G lanes: value of lane t-1 (shift_up1) or t+1 (shift_down1),
// `fill` at the group boundary.  One DPP move on the vector pipe, no LDS round trip: a group of 16 is
// one DPP row (row_shr:1 / row_shl:1, the boundary keeps `fill`); wider groups use the whole-wave
// shift of gfx9 (wave_shr:1 / wave_shl:1) and, for G = 32, one select for the lane at the seam.
__device__ __forceinline__ int dpp_row_shr1(int fill, int v) { return __builtin_amdgcn_update_dpp(fill, v, 0x111, 0xF, 0xF, false); }
__device__ __forceinline__ int dpp_row_shl1(int fill, int v) { return __builtin_amdgcn_update_dpp(fill, v, 0x101, 0xF, 0xF, false); }
__device__ __forceinline__ int dpp_wave_shr1(int fill, int v) { return __builtin_amdgcn_update_dpp(fill, v, 0x138, 0xF, 0xF, false); }
__device__ __forceinline__ int dpp_wave_shl1(int fill, int v) { return __builtin_amdgcn_update_dpp(fill, v, 0x130, 0xF, 0xF, false); }

template <int G>
__device__ __forceinline__ int shift_up1(int v, int fill, int t) {
  if constexpr (G == 16) return dpp_row_shr1(fill, v);
  const int r = dpp_wave_shr1(fill, v);
  if constexpr (G == 64) return r;
  return t == 0 ? fill : r;
}
template <int G>
__device__ __forceinline__ float shift_up1(float v, float fill, int t) {
  return __int_as_float(shift_up1<G>(__float_as_int(v), __float_as_int(fill), t));
}
template <int G>
__device__ __forceinline__ int shift_down1(int v, int fill, int t) {
  if constexpr (G == 16) return dpp_row_shl1(fill, v);
  const int r = dpp_wave_shl1(fill, v);
  if constexpr (G == 64) return r;
  return t == G - 1 ? fill : r;
}
template <int G>
__device__ __forceinline__ float shift_down1(float v, float fill, int t) {
  return __int_as_float(shift_down1<G>(__float_as_int(v), __float_as_int(fill), t));
}

// Columns per lane for this wavefront: W, or W-1 when every pair of the wave fits (wave-uniform, in an SGPR).
template <int G, int W>
__device__ __forceinline__ int pair_width(int L2) {
  if constexpr (W == 1) return 1;
  int m = L2;
#pragma unroll
  for (int o = G; o < 64; o <<= 1) m = max(m, __shfl_xor(m, o));
  m = __builtin_amdgcn_readfirstlane(m);
  return (m + 1 <= G * (W - 1)) ? W - 1 : W;
}

template <int G, int W, class Args, class Post>
__device__ __forceinline__ void pair_finish(const Args& a, float* __restrict__ slab, uint32_t* __restrict__ s_rowptr, int lane, int t, int g,
                                            int L1, int L2, int nsteps, int wr, bool act, uint32_t task, float th, Post post) {
  const bool full = wr == W;
  const int tlast = (L2 >= 0 ? L2 : 0) / wr;  // lane (within the group) that owns column L2
  const int j0 = t * wr;
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
      float* __restrict__ slab_s = slab + (size_t)s * (W * 64);
      const float rdp = shift_up1<G>(lastdp, 0.0f, t);
      const int rtr = shift_up1<G>(lasttr, 0, t), rcnt = shift_up1<G>(lastcnt, 0, t);
      float ddp = dgdp, ldp = rdp;
      int dtr = dgtr, ltr = rtr, run = rcnt;
      float sv[W];
#pragma unroll
      for (int c = 0; c < W; ++c) {
        if (c < W - 1 || full) sv[c] = slab_s[c * 64 + lane];  // private slot: unguarded (cells outside the grid are ignored below)
      }
#pragma unroll
      for (int c = 0; c < W; ++c) {
        if (c < W - 1 || full) {
          const int j = j0 + c;
          const bool v = rowv && (j <= L2);
          const bool inner = v && i >= 1 && j >= 1;
          // the model's posterior; wrapper (>= th keeps) then adapter (> th keeps): align.cpp:69-78
          const float p = post(sv[c]);
          const bool entry = inner && (p >= th) && (p > th);
          slab_s[c * 64 + lane] = entry ? p : 0.0f;
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
        }
      }
      dgdp = rdp; dgtr = rtr;
      lastdp = ldp; lasttr = ltr; lastcnt = run;
      if (i == L1 && t == tlast) {  // dafs.cpp:763; one lane of the group, once
        const int cl = L2 - j0;
#pragma unroll
        for (int c = 0; c < W; ++c)
          if (c == cl) simv = pdp[c] / (float)ptr[c];
      }
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
      const int j = j0 + c;
      if ((c < W - 1 || full) && j >= 1 && j <= L2) a.rowptr_pool[rp + L1 + 1 + j] = (uint32_t)(colbase[c] + colcnt[c]);
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
      const float* __restrict__ slab_s = slab + (size_t)s * (W * 64);
      int run = shift_up1<G>(lastcnt, 0, t);
      const uint32_t rowbase = (rowv && i >= 1) ? s_rowptr[i - 1] : 0;
      float pv[W];
#pragma unroll
      for (int c = 0; c < W; ++c) {
        if (c < W - 1 || full) pv[c] = slab_s[c * 64 + lane];  // sweep 3 left 0 in every non-entry slot
      }
#pragma unroll
      for (int c = 0; c < W; ++c) {
        if (c < W - 1 || full) {
          const int j = j0 + c;
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
      }
      lastcnt = run;
    }
  }
  wave_lds_fence();
}

// ---------------------------------------------------------------------------------------------
// host side: choice of the kernel instance for a batch
// ---------------------------------------------------------------------------------------------
struct pair_variant {
  int G, W;
  const void* fn;   // kernel entry (hipFuncGetAttributes / launch)
  int vgprs;        // registers per lane of the code object (0 until queried)
};

// waves per SIMD the register file allows (MI355X_MICROARCH.md, register files: 512 per lane per SIMD, granule 8)
inline int pair_occupancy(int vgprs) {
  const int alloc = (vgprs + 7) / 8 * 8;
  const int w = 512 / (alloc < 64 ? 64 : alloc);
  return w < 1 ? 1 : w;
}

// Picks (G, W) and the number of persistent wavefronts.  The model: a wavefront executes
// (max_len1 + G) steps of (step_cost + W * cell_cost) instructions; a SIMD that holds `occ` wavefronts at
// once issues one vector instruction every 2 cycles when enough of them are ready and a lone wavefront one
// every ~12 (its own issue rate plus the LDS-lookup latency in every log-sum-exp), so a SIMD with n wavefronts
// to run needs  work * max(12 * ceil(n / occ), 2.6 * n)  cycles.  occ comes from the code object's register
// count (hipFuncGetAttributes), not from a guess.
inline int pair_choose(pair_variant* vs, int nv, uint32_t ntasks, uint32_t max_len1, uint32_t max_len2, int planes,
                       double step_cost, double cell_cost, dafs_pairhmm_plan* plan) {
  int cus = 256, dev = 0;
  if (hipGetDevice(&dev) == hipSuccess) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
  } else {
    (void)hipGetLastError();
  }
  const char* e;
  const int force_g = (e = getenv("DAFS_HIP_FORCE_GROUP")) ? atoi(e) : 0;    // tests exercise every group size
  const int force_w = (e = getenv("DAFS_HIP_FORCE_WIDTH")) ? atoi(e) : 0;    // tuning
  const int cap_occ = (e = getenv("DAFS_HIP_WAVES_PER_SIMD")) ? atoi(e) : 0; // tuning: fewer resident wavefronts than the registers allow
  const pair_variant* best = nullptr;
  double best_cost = 0;
  int best_occ = 1;
  for (int k = 0; k < nv; ++k) {
    pair_variant& v = vs[k];
    if ((uint64_t)v.G * v.W < (uint64_t)max_len2 + 1) continue;
    if (force_g && v.G != force_g) continue;
    if (force_w && v.W != force_w) continue;
    if (v.vgprs == 0) {
      hipFuncAttributes at;
      if (hipFuncGetAttributes(&at, v.fn) == hipSuccess && at.numRegs > 0) v.vgprs = at.numRegs;
      else { (void)hipGetLastError(); v.vgprs = 88 + 8 * v.W; }
    }
    int occ = pair_occupancy(v.vgprs);
    if (cap_occ > 0 && occ > cap_occ) occ = cap_occ;
    const uint64_t waves = ((uint64_t)ntasks + (64 / v.G) - 1) / (64 / v.G);
    const double n = (double)waves / (4.0 * cus);
    const double rounds = (double)((waves + (uint64_t)occ * 4 * cus - 1) / ((uint64_t)occ * 4 * cus));
    const double work = (double)(max_len1 + v.G) * (step_cost + v.W * cell_cost);
    const double a = 12.0 * rounds, b = 2.6 * n;
    const double cost = work * (a > b ? a : b);
    if (!best || cost < best_cost) { best = &v; best_cost = cost; best_occ = occ; }
  }
  if (!best) return DAFS_HIP_ETOOLONG;
  const uint64_t waves = ((uint64_t)ntasks + (64 / best->G) - 1) / (64 / best->G);
  const uint64_t resident = (uint64_t)best_occ * 4 * cus;
  plan->group = best->G;
  plan->width = best->W;
  uint32_t nw = (uint32_t)(waves < resident ? waves : resident);
  nw = (nw + 3) & ~3u;  // whole workgroups of 4 waves
  plan->nwaves = nw;
  plan->slab_steps = max_len1 + best->G;
  plan->scratch_bytes = (uint64_t)nw * plan->slab_steps * best->W * 64 * sizeof(float) * planes;
  return DAFS_HIP_OK;
}

}  // namespace dafs
