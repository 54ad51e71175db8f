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
// pairs are short enough runs the W-1 instantiation of every sweep (lane t owns columns
// t*WR .. t*WR+WR-1, WR = W or W-1; the slab keeps stride W).  Lengths inside one batch differ by a few
// per cent (the benchmark sets: +-7 %), and with 5 instead of 6 columns that is 17 % of the cells.
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

// WR = columns this wave's lanes own (W or W-1, compile time here: the kernels branch once per pair on pair_width and
// instantiate every sweep for both), W = slab stride.  post(sv, p) turns the WR slab values of a step into posteriors.
//
// Entries are rare (a few per row), so sweep 3 does not write the thresholded plane back for a fourth sweep over every
// cell to re-read: each lane appends a 12-byte record per entry {p, row-1 | c<<16, position in its row | position in
// its column << 16} to a private list (`list`: a second plane of the wave's scratch, same [k*64 + lane] layout), and
// sweep 4 walks the lists -- a third less HBM traffic for the whole kernel, and ~40 list steps instead of ~190 grid
// steps.  A lane whose list is full (more than a third of its cells are entries: th near 0) makes the call return
// false with nothing emitted and the slab untouched; the caller then runs the dense = true instantiation, which is
// the plane-and-rescan form.
template <int G, int W, int WR, bool dense, class Args, class Post>
__device__ __forceinline__ bool pair_finish(const Args& a, float* __restrict__ slab, float* __restrict__ list, int list_cap, uint32_t* __restrict__ s_rowptr,
                                            int lane, int t, int g, int L1, int L2, int nsteps, bool act, uint32_t task, float th, Post post) {
  const int tlast = (L2 >= 0 ? L2 : 0) / WR;  // lane (within the group) that owns column L2
  const int j0 = t * WR;
  // ------------------------------------------------------------------ sweep 3: posterior + sim + counts
  int colcnt[WR];
  float simv = 0.0f;
  uint32_t nnz = 0;
  int slab_nrec = 0;
  {
    float pdp[WR];
    int ptr[WR];
#pragma unroll
    for (int c = 0; c < WR; ++c) { pdp[c] = 0.0f; ptr[c] = 0; colcnt[c] = 0; }
    float lastdp = 0.0f, dgdp = 0.0f;
    int lasttr = 0, dgtr = 0, lastcnt = 0;
    uint32_t rowacc = 0;
    if (t == G - 1) s_rowptr[0] = 0;
    int nrec = 0;       // records in this lane's list
    bool ovf = false;
    // slab values are fetched one step ahead (a load issued where it is needed waits for itself and for the stores of
    // the step before: loads and stores share one in-order counter)
    float sv[WR];
#pragma unroll
    for (int c = 0; c < WR; ++c) sv[c] = slab[c * 64 + lane];
    for (int s = 0; s < nsteps; ++s) {
      const int i = s - t;
      const bool rowv = (i >= 0) && (i <= L1);
      float* __restrict__ slab_s = slab + (size_t)s * (W * 64);
      const float rdp = shift_up1<G>(lastdp, 0.0f, t);
      const int rtr = shift_up1<G>(lasttr, 0, t), rcnt = shift_up1<G>(lastcnt, 0, t);
      float ddp = dgdp, ldp = rdp;
      int dtr = dgtr, ltr = rtr, run = rcnt;
      float pp[WR];
      post(sv, pp);  // the model's posteriors of the WR cells, table reads batched
      {  // private slots: unguarded (cells outside the grid are ignored below); the last step refetches its own
        const float* __restrict__ slab_n = slab_s + (s + 1 < nsteps ? W * 64 : 0);
#pragma unroll
        for (int c = 0; c < WR; ++c) sv[c] = slab_n[c * 64 + lane];
      }
      // One straight-line block for the WR cells (the posteriors' polynomials, the similarity DP and the counts as
      // selects: five independent chains the scheduler can interleave), then the rare appends.  Written as per-cell
      // if / else, every cell becomes a handful of basic blocks and its polynomial waits for the cell before it.
      bool ent[WR];
      int runa[WR];  // entries of the row up to and including cell c
#pragma unroll
      for (int c = 0; c < WR; ++c) {
        const int j = j0 + c;
        const bool v = rowv && (j <= L2);
        const bool inner = v && i >= 1 && j >= 1;
        // wrapper (>= th keeps) then adapter (> th keeps): align.cpp:69-78
        const float p = pp[c];
        const bool entry = inner && (p >= th) && (p > th);
        ent[c] = entry;
        if (dense) slab_s[c * 64 + lane] = entry ? p : 0.0f;
        // calculate_similarity_score, dafs.cpp:720-760: an entry starts from the diagonal (dp = ddp + p) and is replaced
        // by the left, then the upper neighbour where that is strictly larger; a non-entry starts from the left one.
        // Every dp is >= 0, so a start of -1 for a non-entry always takes the left neighbour.
        const float udp = pdp[c];
        const int utr = ptr[c];
        const float dpe = entry ? ddp + p : -1.0f;
        const bool fl = dpe < ldp;
        float dp = fl ? ldp : dpe;
        int tr = fl ? ltr : dtr;
        const bool fu = dp < udp;
        dp = fu ? udp : dp;
        tr = (fu ? utr : tr) + 1;
        if (!inner) { dp = 0.0f; tr = 0; }
        ddp = udp; dtr = utr;
        pdp[c] = dp; ptr[c] = tr;
        ldp = dp; ltr = tr;
        run += entry ? 1 : 0;
        colcnt[c] += entry ? 1 : 0;
        runa[c] = run;
      }
      if (!dense) {
        bool any = false;
#pragma unroll
        for (int c = 0; c < WR; ++c) any = any || ent[c];
        if (any) {
#pragma unroll
          for (int c = 0; c < WR; ++c) {
            if (ent[c]) {
              if (3 * nrec + 3 <= list_cap) {
                float* __restrict__ r = list + (size_t)(3 * nrec) * 64 + lane;
                r[0] = pp[c];
                r[64] = __int_as_float((i - 1) | (c << 16));
                r[128] = __int_as_float((runa[c] - 1) | ((colcnt[c] - 1) << 16));  // positions before this entry
                ++nrec;
              } else {
                ovf = true;
              }
            }
          }
        }
      }
      dgdp = rdp; dgtr = rtr;
      lastdp = ldp; lasttr = ltr; lastcnt = run;
      if (i == L1 && t == tlast) {  // dafs.cpp:763; one lane of the group, once
        const int cl = L2 - j0;
        float sdp = 0.0f;
        int str = 1;
#pragma unroll
        for (int c = 0; c < WR; ++c)
          if (c == cl) { sdp = pdp[c]; str = ptr[c]; }
        simv = sdp / (float)str;
      }
      if (t == G - 1 && rowv && i >= 1) {  // row i is complete: its count has crossed the group
        rowacc += (uint32_t)run;
        s_rowptr[i] = rowacc;
      }
    }
    nnz = rowacc;
    if (!dense && __any(ovf)) return false;  // wave-uniform: every pair of the wave takes the dense form
    slab_nrec = nrec;
  }
  nnz = __shfl(nnz, g * G + (G - 1));
  simv = __shfl(simv, g * G + tlast);

  // column prefix sums (row pointers of the transposed matrix)
  int colbase[WR];
  {
    int mine = 0;
#pragma unroll
    for (int c = 0; c < WR; ++c) mine += colcnt[c];
    int incl = mine;
#pragma unroll
    for (int o = 1; o < G; o <<= 1) {
      const int up = __shfl_up(incl, o, G);
      if (t >= o) incl += up;
    }
    int run = incl - mine;
#pragma unroll
    for (int c = 0; c < WR; ++c) { colbase[c] = run; run += colcnt[c]; }
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
    for (int c = 0; c < WR; ++c) {
      const int j = j0 + c;
      if (j >= 1 && j <= L2) a.rowptr_pool[rp + L1 + 1 + j] = (uint32_t)(colbase[c] + colcnt[c]);
    }
  }

  // ------------------------------------------------------------------ sweep 4: emit CSR + transposed CSR
  if (!dense) {
    if (__any(ok)) {
      int maxrec = slab_nrec;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) maxrec = max(maxrec, __shfl_xor(maxrec, o));
      maxrec = __builtin_amdgcn_readfirstlane(maxrec);
      float rp = 0.0f;
      int r1 = 0, r2 = 0;
      if (slab_nrec > 0) { rp = list[lane]; r1 = __float_as_int(list[64 + lane]); r2 = __float_as_int(list[128 + lane]); }
      for (int k = 0; k < maxrec; ++k) {
        const bool live = k < slab_nrec;
        const float p = rp;
        const int q1 = r1, q2 = r2;
        if (k + 1 < slab_nrec) {  // next record, one iteration ahead
          const float* __restrict__ r = list + (size_t)(3 * (k + 1)) * 64 + lane;
          rp = r[0]; r1 = __float_as_int(r[64]); r2 = __float_as_int(r[128]);
        }
        if (live && ok) {
          const int im1 = q1 & 0xFFFF, c = q1 >> 16;
          int cb = 0;
#pragma unroll
          for (int cc = 0; cc < WR; ++cc) cb = (c == cc) ? colbase[cc] : cb;
          const unsigned long long pos = off + s_rowptr[im1] + (uint32_t)(q2 & 0xFFFF);
          a.ent_col[pos] = (uint32_t)(j0 + c - 1);
          a.ent_val[pos] = p;
          const unsigned long long tpos = off + nnz + (uint32_t)(cb + (q2 >> 16));
          a.ent_col[tpos] = (uint32_t)im1;
          a.ent_val[tpos] = p;
        }
      }
    }
  } else if (__any(ok)) {
    int colrun[WR];
#pragma unroll
    for (int c = 0; c < WR; ++c) colrun[c] = 0;
    int lastcnt = 0;
    float pv[WR];
#pragma unroll
    for (int c = 0; c < WR; ++c) pv[c] = slab[c * 64 + lane];
    for (int s = 0; s < nsteps; ++s) {
      const int i = s - t;
      const bool rowv = (i >= 0) && (i <= L1);
      const float* slab_s = slab + (size_t)s * (W * 64);
      int run = shift_up1<G>(lastcnt, 0, t);
      const uint32_t rowbase = (rowv && i >= 1) ? s_rowptr[i - 1] : 0;
#pragma unroll
      for (int c = 0; c < WR; ++c) {
        const int j = j0 + c;
        const float p = pv[c];  // sweep 3 left 0 in every non-entry slot
        if (s + 1 < nsteps) pv[c] = slab_s[(W + c) * 64 + lane];
        const bool entry = p != 0.0f;
        if (entry && ok) {
          const unsigned long long pos = off + rowbase + (uint32_t)run;
          a.ent_col[pos] = (uint32_t)(j - 1);
          a.ent_val[pos] = p;
          const unsigned long long tpos = off + nnz + (uint32_t)(colbase[c] + colrun[c]);
          a.ent_col[tpos] = (uint32_t)(i - 1);
          a.ent_val[tpos] = p;
        }
        run += entry ? 1 : 0;
        colrun[c] += entry ? 1 : 0;
      }
      lastcnt = run;
    }
  }
  wave_lds_fence();
  return true;
}

// ---------------------------------------------------------------------------------------------
// host side: choice of the kernel instance for a batch
// ---------------------------------------------------------------------------------------------
struct pair_variant {
  int G, W;
  const void* fn;   // kernel entry (hipFuncGetAttributes / launch)
  int vgprs;        // registers per lane of the code object (0 until queried)
  int occ;          // wavefronts per SIMD the instance was compiled for (__launch_bounds__): stands in without a device
};

// waves per SIMD the register file allows (MI355X_MICROARCH.md, register files: 512 per lane per SIMD, granule 8)
inline int pair_occupancy(int vgprs) {
  const int alloc = (vgprs + 7) / 8 * 8;
  const int w = 512 / (alloc < 64 ? 64 : alloc);
  return w < 1 ? 1 : w;
}

// Picks (G, W) and the number of persistent wavefronts.  The model, fitted to runs of the ProbCons kernel on MI355X
// (profiles/r02_b_variants.txt): a wavefront executes (max_len1 + G) steps, each worth step_cost + cell_cost * (W - 1/2)
// ns of its SIMD's issue time (most waves of a batch run the W-1 instantiation); a SIMD that holds k wavefronts at
// once works at eff(k) = 0.5 / 0.75 / 1 of its issue rate for k = 1 / 2 / 3 or more (the sweeps wait on LDS lookups
// and on their slab); whole-wave groups spend whole_wave_factor of that per step (ProbCons 0.83: no seam, uniform
// pair bounds; CONTRAlign 1); and when the batch needs r > 1 rounds of resident wavefronts the rounds overlap each
// other's memory-bound and issue-bound sweeps (x 1 - round_overlap (1 - 1/r); ProbCons 0.17, CONTRAlign 0: its
// 1.8 us per cell and wavefront hold for every variant measured).  k comes from the code object's register count (hipFuncGetAttributes), not from a guess.
inline int pair_choose(pair_variant* vs, int nv, uint32_t ntasks, uint32_t max_len1, uint32_t max_len2, int planes,
                       double step_cost, double cell_cost, double whole_wave_factor, double round_overlap, dafs_pairhmm_plan* plan) {
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
      else { (void)hipGetLastError(); v.vgprs = 512 / v.occ / 8 * 8; }
    }
    int occ = pair_occupancy(v.vgprs);
    if (cap_occ > 0 && occ > cap_occ) occ = cap_occ;
    const uint64_t waves = ((uint64_t)ntasks + (64 / v.G) - 1) / (64 / v.G);
    const double n = (double)waves / (4.0 * cus);  // wavefronts per SIMD over the whole batch
    const double kres = n < 1.0 ? 1.0 : (n < occ ? n : (double)occ);  // resident at once
    const double eff = kres < 1.5 ? 0.5 : (kres < 2.5 ? 0.75 : 1.0);
    const double step = (step_cost + cell_cost * (v.W - 0.5)) * (v.G == 64 ? whole_wave_factor : 1.0);
    const double rounds = n > occ ? n / occ : 1.0;
    const double cost = (double)(max_len1 + v.G) * step * (n < 1.0 ? 1.0 : n) / eff * (1.0 - round_overlap * (1.0 - 1.0 / rounds));
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
