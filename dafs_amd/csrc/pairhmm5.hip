// dafs_amd/csrc/pairhmm5.hip -- batched CONTRAlign pair-CRF posteriors for gfx950.
//
// Replaces, per sequence pair: CONTRALIGN::InferenceEngine<float>::{ComputeForward,
// ComputeBackward, ComputePosterior} (reference src/contralign/InferenceEngine.ipp:999-1070,
// 1079-1150, 1279-1317; 5 states MATCH, INS_X, INS_Y, INS2_X, INS2_Y; 24 RNA weights
// src/contralign/Defaults.ipp:389-419), CONTRAlign::calculate's dense->sparse step
// (src/align.cpp:87-106), transpose_mp and calculate_similarity_score (src/dafs.cpp:155-167,713-764).
//
// Same machine mapping as pairhmm3.hip (G lanes per pair, W columns per lane, rows skewed by
// lane, boundary column handed to the neighbour lane by shuffle).  Differences:
//   * the posterior needs, per cell, the five forward terms a_k = Ff[k](i-1,j-1) + ScoreMatch(i,j,k)
//     and only Fb[MATCH](i,j); the normaliser Z is the FORWARD partition function, known when the
//     forward sweep ends.  So sweep 1 stores five planes a_k, and sweep 2 (backward) turns them
//     into the clipped posterior in plane 0 on the fly; sweeps 3-4 are the shared pair_finish.
//   * the reference's backward pass scatters; every cell here gathers its addends in the order
//     that scatter delivers them (diagonal source, then the source below, then the source to the
//     right -- derivation in DESIGN.md), so the log-sum-exp chains round identically.  Row 0 and
//     column 0 of the backward tables feed nothing the posterior reads and are not computed.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "../../include/dafs_hip.h"
#include "contra_math.h"
#include "hip_util.h"
#include "stage.h"
#include "pair_sweeps.h"

namespace dafs {

enum { cM = 0, cIX = 1, cIY = 2, cI2X = 3, cI2Y = 4 };

// One pair per group of G lanes, all sweeps, for the compile-time column count WR (W or W-1).
template <int G, int W, int WR>
__device__ __forceinline__ void pairhmm5_pair(const dafs_pairhmm5_args& a, const contra_tables* ct, const float* s_match, const float* s_insert,
                                              float* __restrict__ slab, size_t plane, uint32_t* __restrict__ s_rowptr, int lane, int t, int g,
                                              int L1, int L2, int nsteps, bool act, uint32_t task,
                                              const uint8_t* __restrict__ s1, const uint8_t* __restrict__ s2) {
  const float NI = CONTRA_NEG_INF;
  const float th = a.th;
  // transition scores into each state, by source state (InferenceEngine.ipp:139-226): read from the kernel
  // arguments, so they live in scalar registers (from LDS they would take twenty-two vector registers)
  const float (*pr)[5] = a.model.pair;
  const float pMM = pr[cM][cM], pXM = pr[cIX][cM], pYM = pr[cIY][cM], p2XM = pr[cI2X][cM], p2YM = pr[cI2Y][cM];
  const float pMX = pr[cM][cIX], pXX = pr[cIX][cIX], pYX = pr[cIY][cIX];
  const float pMY = pr[cM][cIY], pXY = pr[cIX][cIY], pYY = pr[cIY][cIY];
  const float pM2X = pr[cM][cI2X], p2X2X = pr[cI2X][cI2X], p2Y2X = pr[cI2Y][cI2X];
  const float pM2Y = pr[cM][cI2Y], p2X2Y = pr[cI2X][cI2Y], p2Y2Y = pr[cI2Y][cI2Y];
  const float sgM = a.model.single[cM], sgX = a.model.single[cIX], sgY = a.model.single[cIY], sg2X = a.model.single[cI2X], sg2Y = a.model.single[cI2Y];
  const int j0 = t * WR;
  const int tlast = (L2 >= 0 ? L2 : 0) / WR;

  // y symbols of this lane's columns: cc[c] = y[j], j = t*W + c (CONTRAlign alphabet "ACGU", else 4)
  int cc[WR + 1];
#pragma unroll
  for (int c = 0; c <= WR; ++c) {
    const int j = j0 + c;
    const int code = (j >= 1 && j <= L2) ? (int)s2[j - 1] : 4;
    cc[c] = code < 4 ? code : 4;
  }

  // ------------------------------------------------------------------ sweep 1: forward (:999-1070)
  float Z = NI;
  float eM = NI, eX = NI, eY = NI, e2X = NI, e2Y = NI;
  {
    float pM[WR], pX[WR], pY[WR], p2X[WR], p2Y[WR];  // row i-1 of this lane's columns
#pragma unroll
    for (int c = 0; c < WR; ++c) pM[c] = pX[c] = pY[c] = p2X[c] = p2Y[c] = NI;
    float lsM = NI, lsX = NI, lsY = NI, ls2X = NI, ls2Y = NI;  // last column, row of the previous step
    float dgM = NI, dgX = NI, dgY = NI, dg2X = NI, dg2Y = NI;  // neighbour's last column, one row earlier
    for (int s = 0; s < nsteps; ++s) {
      const int i = s - t;
      const bool rowv = (i >= 0) && (i <= L1);
      int xi = (rowv && i >= 1) ? (int)s1[i - 1] : 4;
      xi = xi < 4 ? xi : 4;
      const float rM = shift_up1<G>(lsM, NI, t), rX = shift_up1<G>(lsX, NI, t), rY = shift_up1<G>(lsY, NI, t);
      const float r2X = shift_up1<G>(ls2X, NI, t), r2Y = shift_up1<G>(ls2Y, NI, t);
      float dM = dgM, dX = dgX, dY = dgY, d2X = dg2X, d2Y = dg2Y;  // (i-1, j-1)
      float lM = rM, lX = rX, lY = rY, l2X = r2X, l2Y = r2Y;       // (i, j-1)
      const float insx = 0.0f + s_insert[xi];
      const float bix = insx + sgX, bi2x = insx + sg2X;            // ScoreInsertX / ScoreInsert2X without the pair term
#pragma unroll
      for (int c = 0; c < WR; ++c) {
        const int j = j0 + c;
        const bool v = rowv && (j <= L2);
        const int yj = cc[c];
        const float insy = 0.0f + s_insert[yj];
        const float biy = insy + sgY, bi2y = insy + sg2Y;
        const float bm = (0.0f + s_match[xi * 5 + yj]) + sgM;
        const bool first = (i == 1 && j == 1);
        // a_k = Ff[k](i-1,j-1) + ScoreMatch(i,j,k), :446-482 (no pair score on the very first match)
        const float aM = dM + (bm + (first ? 0.0f : pMM));
        const float aX = dX + (bm + (first ? 0.0f : pXM));
        const float aY = dY + (bm + (first ? 0.0f : pYM));
        const float a2X = d2X + (bm + (first ? 0.0f : p2XM));
        const float a2Y = d2Y + (bm + (first ? 0.0f : p2YM));
        // Seventeen Fast_LogPlusEquals per cell in the reference; the first of every chain has NEG_INF on the
        // left and returns its argument unchanged (the lo > NEG_INF/2 test fails), the other twelve run as
        // six packed pairs (contra_lpe2_t).  m: aM (+) aX (+) aY (+) a2X (+) a2Y; x, y, x2, y2: three terms each.
        typedef contra_f2 f2;
        const f2 r1 = contra_lpe2_t(ct, f2{aM, pM[c] + (bix + pMX)}, f2{aX, pX[c] + (bix + pXX)});
        const f2 r2 = contra_lpe2_t(ct, f2{r1.x, r1.y}, f2{aY, pY[c] + (bix + pYX)});
        const f2 r3 = contra_lpe2_t(ct, f2{r2.x, lM + (biy + pMY)}, f2{a2X, lX + (biy + pXY)});
        const f2 r4 = contra_lpe2_t(ct, f2{r3.x, r3.y}, f2{a2Y, lY + (biy + pYY)});
        const f2 r5 = contra_lpe2_t(ct, f2{pM[c] + (bi2x + pM2X), lM + (bi2y + pM2Y)}, f2{p2X[c] + (bi2x + p2X2X), l2X + (bi2y + p2X2Y)});
        const f2 r6 = contra_lpe2_t(ct, f2{r5.x, r5.y}, f2{p2Y[c] + (bi2x + p2Y2X), l2Y + (bi2y + p2Y2Y)});
        float m = first ? aM : r4.x;
        float x = r2.y;
        float y = r4.y;
        float x2 = r6.x;
        float y2 = r6.y;
        if (i == 0 || j == 0) {  // borders, :1005-1010: only the insert chains run along row 0 / column 0
          m = NI; x = NI; y = NI; x2 = NI; y2 = NI;
          if (i == 0 && j == 0) { m = 0.0f; x = 0.0f; y = 0.0f; x2 = 0.0f; y2 = 0.0f; }
          else if (i == 0) {  // NEG_INF (+) v == v
            y = lY + (biy + (j != 1 ? pYY : 0.0f));
            y2 = l2Y + (bi2y + (j != 1 ? p2Y2Y : 0.0f));
          } else {
            x = pX[c] + (bix + (i != 1 ? pXX : 0.0f));
            x2 = p2X[c] + (bi2x + (i != 1 ? p2X2X : 0.0f));
          }
        }
        if (!v) { m = NI; x = NI; y = NI; x2 = NI; y2 = NI; }
        dM = pM[c]; dX = pX[c]; dY = pY[c]; d2X = p2X[c]; d2Y = p2Y[c];
        pM[c] = m; pX[c] = x; pY[c] = y; p2X[c] = x2; p2Y[c] = y2;
        lM = m; lX = x; lY = y; l2X = x2; l2Y = y2;
        if (v && i >= 1 && j >= 1) {
          float* __restrict__ q = slab + (size_t)s * (W * 64) + (c * 64 + lane);
          q[0] = aM; q[plane] = aX; q[2 * plane] = aY; q[3 * plane] = a2X; q[4 * plane] = a2Y;
        }
      }
      dgM = rM; dgX = rX; dgY = rY; dg2X = r2X; dg2Y = r2Y;
      lsM = lM; lsX = lX; lsY = lY; ls2X = l2X; ls2Y = l2Y;
      if (i == L1 && t == tlast) {  // the five F_k(L1, L2): one lane of the group, once
        const int cl = L2 - j0;
#pragma unroll
        for (int c = 0; c < WR; ++c)
          if (c == cl) { eM = pM[c]; eX = pX[c]; eY = pY[c]; e2X = p2X[c]; e2Y = p2Y[c]; }
      }
    }
  }
  {  // ComputeForwardLogPartitionCoefficient, :1164-1170
    float z = eM;
    z = contra_lpe(z, eX); z = contra_lpe(z, eY); z = contra_lpe(z, e2X); z = contra_lpe(z, e2Y);
    Z = __shfl(z, g * G + tlast);
  }

  // ------------------------------------------------------------------ sweep 2: backward (:1079-1150) + posterior (:1279-1317)
  {
    float pM[WR], pX[WR], p2X[WR];      // row a+1 of this lane's columns: Fb[M], Fb[IX], Fb[I2X]
#pragma unroll
    for (int c = 0; c < WR; ++c) pM[c] = pX[c] = p2X[c] = NI;
    float fsM = NI, fsY = NI, fs2Y = NI;  // this lane's first column, row of the previous step
    float dgM = NI;                       // right neighbour's first column, one row later
    for (int s = 0; s < nsteps; ++s) {
      const int sf = nsteps - 1 - s;  // forward step of row i for this lane: wave-uniform (pairhmm3.hip, sweep 2)
      const int i = sf - t;
      const bool rowv = (i >= 1) && (i <= L1);
      float* __restrict__ slab_s = slab + (size_t)sf * (W * 64);
      int xn = (rowv && i < L1) ? (int)s1[i] : 4;  // x[i+1]
      xn = xn < 4 ? xn : 4;
      const float rM = shift_down1<G>(fsM, NI, t), rY = shift_down1<G>(fsY, NI, t), r2Y = shift_down1<G>(fs2Y, NI, t);
      float dM = dgM;              // Fb[M](i+1, j+1)
      float gY = rY, g2Y = r2Y;    // Fb[IY](i, j+1), Fb[I2Y](i, j+1)
      const float insx = 0.0f + s_insert[xn];
      const float bix = insx + sgX, bi2x = insx + sg2X;
      float ak[5][WR];
#pragma unroll
      for (int c = 0; c < WR; ++c) {
        const int j = j0 + c;
        const bool v = rowv && j >= 1 && j <= L2;
#pragma unroll
        for (int k = 0; k < 5; ++k) ak[k][c] = v ? slab_s[k * plane + (c * 64 + lane)] : 0.0f;
      }
#pragma unroll
      for (int c = WR - 1; c >= 0; --c) {
        const int j = j0 + c;
        const bool v = rowv && j >= 1 && j <= L2;
        const int yn = cc[c + 1];  // y[j+1]
        const float insy = 0.0f + s_insert[yn];
        const float biy = insy + sgY, bi2y = insy + sg2Y;
        const float bm = (0.0f + s_match[xn * 5 + yn]) + sgM;
        // sources in delivery order: (i+1,j+1) match block, (i+1,j) insert-X blocks, (i,j+1) insert-Y blocks
        // the first addend of every state arrives on NEG_INF and is taken as it is; the other twelve
        // Fast_LogPlusEquals run as six packed pairs, in delivery order per state
        typedef contra_f2 f2;
        const float vX = pX[c], v2X = p2X[c];
        const f2 q1 = contra_lpe2_t(ct, f2{dM + (bm + pMM), dM + (bm + pXM)}, f2{vX + (bix + pMX), vX + (bix + pXX)});        // bM, bX
        const f2 q2 = contra_lpe2_t(ct, f2{dM + (bm + pYM), dM + (bm + p2XM)}, f2{vX + (bix + pYX), v2X + (bi2x + p2X2X)});    // bY, b2X
        const f2 q3 = contra_lpe2_t(ct, f2{q1.x, dM + (bm + p2YM)}, f2{v2X + (bi2x + pM2X), v2X + (bi2x + p2Y2X)});            // bM, b2Y
        const f2 q4 = contra_lpe2_t(ct, f2{q3.x, q1.y}, f2{gY + (biy + pMY), gY + (biy + pXY)});                                // bM, bX
        const f2 q5 = contra_lpe2_t(ct, f2{q2.x, q4.x}, f2{gY + (biy + pYY), g2Y + (bi2y + pM2Y)});                             // bY, bM
        const f2 q6 = contra_lpe2_t(ct, f2{q2.y, q3.y}, f2{g2Y + (bi2y + p2X2Y), g2Y + (bi2y + p2Y2Y)});                        // b2X, b2Y
        float bM = q5.y, bX = q4.y, bY = q5.x, b2X = q6.x, b2Y = q6.y;
        if (i == L1 && j == L2) { bM = 0.0f; bX = 0.0f; bY = 0.0f; b2X = 0.0f; b2Y = 0.0f; }  // :1084
        if (!v) { bM = NI; bX = NI; bY = NI; b2X = NI; b2Y = NI; }
        dM = pM[c];
        pM[c] = bM; pX[c] = bX; p2X[c] = b2X;
        gY = bY; g2Y = b2Y;
        if (v) {  // ComputePosterior :1289-1305 + Clip :1308-1315
          float p = 0.0f;
          const contra_f2 e01 = contra_exp2_t(ct, contra_f2{ak[0][c] + bM - Z, ak[1][c] + bM - Z});
          const contra_f2 e23 = contra_exp2_t(ct, contra_f2{ak[2][c] + bM - Z, ak[3][c] + bM - Z});
          const contra_f2 e4 = contra_exp2_t(ct, contra_f2{ak[4][c] + bM - Z, -20.0f});
          p += e01.x;
          if (i > 1 || j > 1) {
            p += e01.y;
            p += e23.x;
            p += e23.y;
            p += e4.x;
          }
          const float mx = p < 0.0f ? 0.0f : p;
          slab_s[c * 64 + lane] = (1.0f < mx) ? 1.0f : mx;
        }
        if (c == 0) { fsY = bY; fs2Y = b2Y; }
      }
      dgM = rM;
      fsM = pM[0];
    }
  }
  // row 0 / column 0 of plane 0 were never written: pair_finish ignores them (inner cells only)
  // row 0 / column 0 of plane 0 were never written: pair_finish ignores them (inner cells only)
  auto post = [](const float (&sv)[WR], float (&p)[WR]) {
#pragma unroll
    for (int c = 0; c < WR; ++c) p[c] = sv[c];
  };
  // sparse outputs through per-lane entry lists (pair_sweeps.h) kept in plane 1, dead since sweep 2; with th near 0
  // (dense outputs), or when a list is full, the plane-and-rescan form
  float* __restrict__ list = slab + plane;
  const int list_cap = (int)(plane / 64);
  if (th < 0.002f || !pair_finish<G, W, WR, false>(a, slab, list, list_cap, s_rowptr, lane, t, g, L1, L2, nsteps, act, task, th, post))
    (void)pair_finish<G, W, WR, true>(a, slab, list, list_cap, s_rowptr, lane, t, g, L1, L2, nsteps, act, task, th, post);
}

template <int G, int W, int OCC>
__global__ __launch_bounds__(256, OCC) void k_pairhmm5(dafs_pairhmm5_args a, uint32_t slab_steps, uint32_t rp_cap) {
  constexpr int NG = 64 / G;
  extern __shared__ uint32_t s_dyn[];
  __shared__ float s_match[25], s_insert[5];
  __shared__ contra_tables s_ct;
  contra_tables_init(&s_ct, threadIdx.x);
  if (threadIdx.x < 25) s_match[threadIdx.x] = (&a.model.match[0][0])[threadIdx.x];
  if (threadIdx.x < 5) s_insert[threadIdx.x] = a.model.insert[threadIdx.x];
  __syncthreads();

  const int lane = threadIdx.x & 63;
  const int t = lane % G;
  const int g = lane / G;
  const int wave_in_wg = threadIdx.x >> 6;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + wave_in_wg);  // scalar slab base, see pairhmm3.hip
  const size_t plane = (size_t)slab_steps * W * 64;
  float* __restrict__ slab = a.scratch + (size_t)wave * plane * 5;
  uint32_t* __restrict__ s_rowptr = s_dyn + (size_t)(wave_in_wg * NG + g) * rp_cap;

  for (;;) {
    uint32_t base = 0;
    if (lane == 0) base = atomicAdd(a.queue, (uint32_t)NG);
    base = __builtin_amdgcn_readfirstlane(base);
    if (base >= a.ntasks) break;
    const uint32_t task = base + g;
    const bool act = task < a.ntasks;
    dafs_pair_task tk = {0, 0, 0, 0};
    if (act) tk = a.tasks[task];
    const int L1 = act ? (int)tk.len1 : -1;
    const int L2 = act ? (int)tk.len2 : -1;
    const uint8_t* __restrict__ s1 = a.codes + tk.off1;
    const uint8_t* __restrict__ s2 = a.codes + tk.off2;
    int maxL1 = L1;
#pragma unroll
    for (int o = G; o < 64; o <<= 1) maxL1 = max(maxL1, __shfl_xor(maxL1, o));
    const int nsteps = __builtin_amdgcn_readfirstlane(maxL1) + G;
    const bool full = pair_width<G, W>(L2) == W;  // columns per lane of this wave: W or W-1 (pair_sweeps.h)
    if (full)
      pairhmm5_pair<G, W, W>(a, &s_ct, s_match, s_insert, slab, plane, s_rowptr, lane, t, g, L1, L2, nsteps, act, task, s1, s2);
    else
      pairhmm5_pair<G, W, (W > 1 ? W - 1 : 1)>(a, &s_ct, s_match, s_insert, slab, plane, s_rowptr, lane, t, g, L1, L2, nsteps, act, task, s1, s2);
  }
}

#define V(G, W, OCC) {G, W, (const void*)k_pairhmm5<G, W, OCC>, 0, OCC}
#ifdef PAIR_ONLY_VARIANTS  // tuning builds: -DPAIR_ONLY_VARIANTS="V(32,6,4),V(64,3,4)" compiles in seconds
static pair_variant k_variants5[] = {PAIR_ONLY_VARIANTS};
#else
static pair_variant k_variants5[] = {
    V(16, 2, 2), V(16, 3, 2), V(16, 4, 2), V(16, 5, 2), V(16, 6, 2), V(16, 8, 2), V(16, 10, 1), V(16, 11, 1), V(16, 12, 1),
    V(32, 2, 2), V(32, 3, 2), V(32, 4, 2), V(32, 5, 2), V(32, 6, 2), V(32, 7, 2), V(32, 8, 2), V(32, 10, 1), V(32, 12, 1),
    V(64, 1, 2), V(64, 2, 2), V(64, 3, 2), V(64, 4, 2), V(64, 5, 2), V(64, 6, 2), V(64, 7, 2), V(64, 8, 2), V(64, 10, 1), V(64, 12, 1), V(64, 16, 1),
};
#endif
#undef V
static const int k_nvariants5 = (int)(sizeof k_variants5 / sizeof k_variants5[0]);

}  // namespace dafs

using namespace dafs;

extern "C" int dafs_hipk_pairhmm5_plan(uint32_t ntasks, uint32_t max_len1, uint32_t max_len2, dafs_pairhmm_plan* plan) {
  if (!plan || ntasks == 0 || max_len1 == 0 || max_len2 == 0) return DAFS_HIP_EINVAL;
  return pair_choose(k_variants5, k_nvariants5, ntasks, max_len1, max_len2, 5, 0.0, 1800.0, 1.0, 0.0, plan);  // five planes
}

extern "C" int dafs_hipk_pairhmm5_launch(const dafs_pairhmm5_args* args, const dafs_pairhmm_plan* plan, void* hip_stream) {
  if (!args || !plan) return DAFS_HIP_EINVAL;
  if (args->ntasks == 0) return DAFS_HIP_OK;
  const pair_variant* v = nullptr;
  for (const pair_variant& c : k_variants5)
    if (c.G == (int)plan->group && c.W == (int)plan->width) v = &c;
  if (!v || plan->nwaves % 4) return DAFS_HIP_EINVAL;
  const uint32_t rp_cap = plan->slab_steps - plan->group + 1;
  const size_t lds = (size_t)4 * (64 / v->G) * rp_cap * sizeof(uint32_t);
  if (lds > 60 * 1024) return DAFS_HIP_ETOOLONG;
  dafs_pairhmm5_args a = *args;
  uint32_t steps = plan->slab_steps, cap = rp_cap;
  void* params[] = {&a, &steps, &cap};
  hipError_t launch_err = hipSuccess;
  STAGE_LAUNCH(dafs::ST_PAIRHMM5, (hipStream_t)hip_stream) launch_err = hipLaunchKernel(v->fn, dim3(plan->nwaves / 4), dim3(256), params, lds, (hipStream_t)hip_stream);
  if (hip_check(launch_err)) return DAFS_HIP_ELAUNCH;
  return DAFS_HIP_OK;
}
