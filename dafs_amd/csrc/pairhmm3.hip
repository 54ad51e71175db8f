// dafs_amd/csrc/pairhmm3.hip -- batched ProbCons pair-HMM posteriors for gfx950.
//
// What it computes, per sequence pair (x,y) (reference file:line it replaces):
//   forward / backward / posterior of the 3-state pair HMM   src/probconsRNA/ProbabilisticModel.h:105-403
//   threshold + dense->sparse rows (p > th)                   src/probconsRNA/wrapper.cpp:125-128, src/align.cpp:69-78
//   the transposed row lists mp[y][x]                         src/dafs.cpp:155-167
//   the similarity score sim[x][y]                            src/dafs.cpp:713-764
//
// Mapping to the machine.  A group of G lanes (G = 16/32/64, so 4/2/1 pairs per wavefront)
// owns one pair.  Lane t of the group owns the W consecutive columns j = t*W .. t*W+W-1 of the
// (L1+1) x (L2+1) DP grid and keeps the previous row of its columns in registers; rows are
// skewed so that at wavefront step s lane t works on row i = s - t (a register-resident
// anti-diagonal wavefront: no LDS traffic and no barriers, the only cross-lane exchange is one
// lane-to-neighbour shuffle of the boundary column per step).  Four sweeps share that shape:
//   1 forward   : writes F_M(i,j) to the wave's slab in HBM
//   2 backward  : mirrored skew; reads F_M, writes S = F_M + B_M in place
//   3 posterior : P = EXP(min(0,S-total)) thresholded, written in place; the same sweep runs the
//                 similarity-score DP and counts entries per row (count travels with the row
//                 from lane to lane) and per column (registers)
//   4 emit      : scatters the entries into the CSR of mp[x][y] and of mp[y][x]
// The slab is indexed [(step*W + c)*64 + lane], so every slab access of a wave instruction is one
// contiguous 256-byte segment, in all four sweeps (the backward sweep visits forward step
// L1+G-1-s, the same for every lane).  Work is handed out by a device-side counter, longest
// pairs first.
//
// Arithmetic is the reference's, operation for operation (pc_math.h); compile with
// -ffp-contract=off.  Results are bit-identical to the CPU path.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "../../include/dafs_hip.h"
#include "pc_math.h"
#include "hip_util.h"
#include "stage.h"
#include "pair_sweeps.h"

namespace dafs {

// One pair per group of G lanes, all four sweeps, for the compile-time column count WR (W or W-1).
template <int G, int W, int WR>
__device__ __forceinline__ void pairhmm3_pair(const dafs_pairhmm3_args& a, const pc_tables* tab, const float* s_match, const float* s_ins, const float2* s_mi,
                                              float* __restrict__ slab, float* __restrict__ list, int list_cap, uint32_t* __restrict__ s_rowptr, int lane, int t, int g,
                                              int L1, int L2, int nsteps, bool act, uint32_t task,
                                              const uint8_t* __restrict__ s1, const uint8_t* __restrict__ s2) {
  // everything below lives in the 32x domain of pc_log_add_n (pc_math.h): the launch scaled the model tables
  const float LZ = PC_LOG_ZERO_S;
  const float i0 = a.model.init[0], i1 = a.model.init[1], i2 = a.model.init[2];
  const float tMM = a.model.trans[0][0], tMX = a.model.trans[0][1], tMY = a.model.trans[0][2];
  const float tXM = a.model.trans[1][0], tXX = a.model.trans[1][1];
  const float tYM = a.model.trans[2][0], tYY = a.model.trans[2][2];
  const int j0 = t * WR;
  const int tlast = (L2 >= 0 ? L2 : 0) / WR;  // lane (within group) that owns column L2

  // residue classes of this lane's columns: cc[c] = class of column j = j0 + c (s2[j-1]); 6 = other.  s_mi[c1*8 + c2]
  // = {match(c1, c2), ins(c2)}: one 8-byte lookup per cell gives the pair emission and the column's single emission
  int cc[WR + 1];
#pragma unroll
  for (int c = 0; c <= WR; ++c) {
    const int j = j0 + c;
    cc[c] = (j >= 1 && j <= L2) ? (int)s2[j - 1] : 6;
  }
  const int c1first = act ? (int)s1[0] : 6;
  const int c2first = act ? (int)s2[0] : 6;
  // the three initial cells (ProbabilisticModel.h:123-131)
  const float fM11 = i0 + s_match[c1first * 8 + c2first];
  const float fX10 = i1 + s_ins[c1first];
  const float fY01 = i2 + s_ins[c2first];

  // Cells outside the grid need no masking: every table starts at LOG_ZERO and the recursions map
  // LOG_ZERO inputs to LOG_ZERO exactly (x + finite == x at 2e20; LOG_ADD returns the other operand
  // when one is LOG_ZERO), so rows a lane has not reached yet and columns beyond L2 stay LOG_ZERO.
  // ------------------------------------------------------------------ sweep 1: forward
  // Per cell (ProbabilisticModel.h:152-155 M, :159-161 X, :165-167 Y) four LOG_ADDs:
  //   m1 = (F_M(i-1,j-1)+tMM) (+) (F_X(i-1,j-1)+tXM)      xx = (F_M(i-1,j)+tMX) (+) (F_X(i-1,j)+tXX)
  //   m2 = m1 (+) (F_Y(i-1,j-1)+tYM)                        yy = (F_M(i,j-1)+tMY) (+) (F_Y(i,j-1)+tYY)
  // Only yy depends on the cell to the left.  The step is a software pipeline over the lane's cells: iteration `it`
  // runs m1 and xx of cell it, m2 of cell it-1 and yy of cell it-2 as one batch of four independent LOG_ADDs
  // (pc_log_add_n: lookups issued together), so a step exposes WR+2 lookup latencies instead of 4*WR.
  float endM = LZ, endX = LZ, endY = LZ;  // F_k(L1, L2)
  {
    float pM[WR], pX[WR], pY[WR];  // row i-1 at this lane's columns
#pragma unroll
    for (int c = 0; c < WR; ++c) pM[c] = pX[c] = pY[c] = LZ;
    float lastM = LZ, lastX = LZ, lastY = LZ;  // this lane's last column, row of the previous step
    float dgM = LZ, dgX = LZ, dgY = LZ;        // neighbour's last column, one row earlier
    // the residue class of a row is fetched one step ahead: a load issued at the top of a step would be waited for at
    // once, together with the previous step's slab stores (loads and stores share one in-order counter)
    int c1n = 6;  // row -t of step 0 is never an inner row
    for (int s = 0; s < nsteps; ++s) {
      const int i = s - t;
      float* __restrict__ slab_s = slab + (size_t)s * (W * 64);
      const int c1 = c1n;
      c1n = (i >= 0 && i < L1) ? (int)s1[i] : 6;  // row i+1 of the next step
      const float rM = shift_up1<G>(lastM, LZ, t), rX = shift_up1<G>(lastX, LZ, t), rY = shift_up1<G>(lastY, LZ, t);
      const float insc1 = s_ins[c1];
      const float2* __restrict__ mi_row = s_mi + c1 * 8;
      float2 mi_prev = make_float2(0.0f, 0.0f), mi_prev2 = mi_prev;  // {match, ins} of cells it-1 and it-2
      // pM/pX/pY are updated in place: a value of row i-1 is overwritten by its row-i successor in the iteration after
      // its last use (M and Y by construction, X one iteration late through xpend)
      float m1prev = LZ, xpend = LZ;
#pragma unroll
      for (int it = 0; it < WR + 2; ++it) {
        float x[4], y[4], o[4];
#pragma unroll
        for (int n = 0; n < 4; ++n) x[n] = y[n] = LZ;
        if (it < WR) {  // cell it: (i-1, j-1) is the neighbour's column for c = 0, else this lane's previous column
          const float dM = it == 0 ? dgM : pM[it > 0 ? it - 1 : 0], dX = it == 0 ? dgX : pX[it > 0 ? it - 1 : 0];
          x[0] = dM + tMM; y[0] = dX + tXM;
          x[1] = pM[it] + tMX; y[1] = pX[it] + tXX;
        }
        if (it >= 1 && it - 1 < WR) {
          const int c = it - 1;
          const float dY = c == 0 ? dgY : pY[c > 0 ? c - 1 : 0];
          x[2] = m1prev; y[2] = dY + tYM;
        }
        if (it >= 2) {
          const int c = it - 2;
          const float lM = c == 0 ? rM : pM[c > 0 ? c - 1 : 0], lY = c == 0 ? rY : pY[c > 0 ? c - 1 : 0];  // row i already
          x[3] = lM + tMY; y[3] = lY + tYY;
        }
        float2 mi_now = make_float2(0.0f, 0.0f);
        if (it < WR) mi_now = mi_row[cc[it]];  // in flight beside the batch's lookups; used one and two iterations on
        pc_log_add_n<4>(tab, x, y, o);
        if (it >= 1 && it - 1 < WR) {
          const int c = it - 1;
          float mv = o[2] + mi_prev.x;
          if (c <= 1 && s <= 2) {  // initial cells, :123-131 (cells with i<=1 && j<=1 are skipped by :150); j <= 1 needs c <= 1,
            const int j = j0 + c;  // and i <= 1 there needs s <= 2: a wave-uniform branch that the other steps skip
            if (i <= 1 && j <= 1) mv = (i == 1 && j == 1) ? fM11 : LZ;
          }
          pM[c] = mv;
          pX[c] = xpend;
#if !defined(PAIR_EXP_NOSTORE)  // tuning experiment: what the slab stores cost
          slab_s[c * 64 + lane] = mv;  // slot (s,c,lane) is private to this lane: no guard needed
#endif
        }
        if (it < WR) {
          m1prev = o[0];
          float xv = insc1 + o[1];
          if (it <= 1 && s <= 2) {
            const int j = j0 + it;
            if (i <= 1 && j <= 1) xv = (i == 1 && j == 0) ? fX10 : LZ;
          }
          xpend = xv;
        }
        if (it >= 2) {
          const int c = it - 2;
          float yv = mi_prev2.y + o[3];
          if (c <= 1 && s <= 2) {
            const int j = j0 + c;
            if (i <= 1 && j <= 1) yv = (i == 0 && j == 1) ? fY01 : LZ;
          }
          pY[c] = yv;
        }
        mi_prev2 = mi_prev;
        mi_prev = mi_now;
      }
      dgM = rM; dgX = rX; dgY = rY;
      lastM = pM[WR - 1]; lastX = pX[WR - 1]; lastY = pY[WR - 1];
      if (i == L1 && t == tlast) {  // F_k(L1, L2): one lane of the group, once
        const int cl = L2 - j0;
#pragma unroll
        for (int c = 0; c < WR; ++c)
          if (c == cl) { endM = pM[c]; endX = pX[c]; endY = pY[c]; }
      }
    }
  }
#if defined(PAIR_EXP_STOP) && PAIR_EXP_STOP == 1  // tuning experiment: time of the forward sweep alone
  if (endM != 12345.0f) return;
#endif
  float totF = LZ;  // ComputeTotalProbability, :341-347 (B_k(L1,L2) = init_k)
  totF = pc_log_add_s(tab, totF, endM + i0);
  totF = pc_log_add_s(tab, totF, endX + i1);
  totF = pc_log_add_s(tab, totF, endY + i2);
  totF = __shfl(totF, g * G + tlast);

  // ------------------------------------------------------------------ sweep 2: backward
  // Per cell (:233-249), with pxy = B_M(i+1,j+1) + match:  the cells start at LOG_ZERO (:213-214 sets the corner to
  // the initial distribution) and LOG_ZERO (+) v == v for every v >= LOG_ZERO (the d >= 7.5 exit; v == LOG_ZERO gives
  // LOG_ZERO), so the first accumulation of each state is a plain assignment; at the corner the right-hand sides
  // are LOG_ZERO (nothing lies beyond it) and the accumulation would return the initial values unchanged.
  //   bm1 = (pxy+tMM) (+) (tx+tMX)   bx = (pxy+tXM) (+) (tx+tXX)          tx = B_X(i+1,j) + ins(x_{i+1})
  //   bm  = bm1 (+) (ty+tMY)         by = (pxy+tYM) (+) (ty+tYY)          ty = B_Y(i,j+1) + ins(y_{j+1})
  // bm and by wait for the cell to the right; same pipeline, columns descending: iteration `it` runs bm1, bx of cell
  // WR-1-it and bm, by of cell WR-it.
  float capM = LZ, capX = LZ, capY = LZ;  // B_M(1,1), B_X(1,0), B_Y(0,1)
  {
    float pM[WR], pX[WR];  // row i+1 at this lane's columns
#pragma unroll
    for (int c = 0; c < WR; ++c) pM[c] = pX[c] = LZ;
    float firstM = LZ, firstY = LZ;  // this lane's first column, row of the previous step
    float dgM = LZ;                  // right neighbour's first column, one row later
    // one step ahead, as in sweep 1: the residue class of the row, and the row's F_M values from the slab (each
    // refetched for the next step as soon as this step has used it, into the same register)
    int c1n;
    float fwd[WR];
    {
      const int i = nsteps - 1 - t;
      const bool rowv = (i >= 0) && (i <= L1);
      c1n = (rowv && i < L1) ? (int)s1[i] : 6;
      const float* __restrict__ q = slab + (size_t)(nsteps - 1) * (W * 64);
#pragma unroll
      for (int c = 0; c < WR; ++c) fwd[c] = (rowv && j0 + c <= L2) ? q[c * 64 + lane] : 0.0f;
    }
    for (int s = 0; s < nsteps; ++s) {
      // mirrored skew: lane G-1 starts with the last row of the longest pair of the wave (rows beyond a shorter
      // pair's L1 stay LOG_ZERO like every other cell outside the grid), so that the forward step that
      // stored row i for this lane, sf = i + t, is the same for the whole wave
      const int sf = nsteps - 1 - s;
      const int i = sf - t;
      const bool rowv = (i >= 0) && (i <= L1);
      float* __restrict__ slab_s = slab + (size_t)sf * (W * 64);
      const int c1 = c1n;
      const int in = i - 1;  // the next step's row; its forward step is sf - 1 (never read when sf == 0: in < 0)
      const bool rowvn = (in >= 0) && (in <= L1);
      c1n = (rowvn && in < L1) ? (int)s1[in] : 6;
      const float rM = shift_down1<G>(firstM, LZ, t), rY = shift_down1<G>(firstY, LZ, t);
      const float insc1 = s_ins[c1];
      const float2* __restrict__ mi_row = s_mi + c1 * 8;
      // pM/pX updated in place (see sweep 1); bm1, by0 and B_Y live for one iteration
      float bm1prev = LZ, by0prev = LZ, yprev = LZ, insprev = 0.0f;
      float2 mi_next = mi_row[cc[WR]];  // {match(x_{i+1}, y_{j+1}), ins(y_{j+1})} of the cell the next iteration starts
#pragma unroll
      for (int it = 0; it < WR + 1; ++it) {
        float x[4], y[4], o[4];
        float by0now = LZ, insnow = 0.0f;
#pragma unroll
        for (int n = 0; n < 4; ++n) x[n] = y[n] = LZ;
        if (it < WR) {
          const int c = WR - 1 - it;
          const int j = j0 + c;
          const float2 mi = mi_next;
          const float dM = c == WR - 1 ? dgM : pM[c < WR - 1 ? c + 1 : 0];  // B_M(i+1, j+1): still row i+1
          const float pxy = dM + mi.x;
          float bm = pxy + tMM, bx = pxy + tXM, by = pxy + tYM;
          if (i == L1 && j == L2) { bm = i0; bx = i1; by = i2; }
          const float tx = pX[c] + insc1;
          x[0] = bm; y[0] = tx + tMX;
          x[1] = bx; y[1] = tx + tXX;
          by0now = by;
          insnow = mi.y;
          if (c >= 1) mi_next = mi_row[cc[c]];  // for cell c-1: class of s2[j-1] (iter2[j]); in flight beside the batch
        }
        if (it >= 1) {
          const int c = WR - it;
          const float rgY = c == WR - 1 ? rY : yprev;  // B_Y(i, j+1)
          const float ty = rgY + insprev;
          x[2] = bm1prev; y[2] = ty + tMY;
          x[3] = by0prev; y[3] = ty + tYY;
        }
        pc_log_add_n<4>(tab, x, y, o);
        if (it >= 1) {
          const int c = WR - it;
          const int j = j0 + c;
          pM[c] = o[2];
          yprev = o[3];
          if (rowv && j <= L2) slab_s[c * 64 + lane] = fwd[c] + o[2];  // forward[ij] + backward[ij], :395
          fwd[c] = (rowvn && j <= L2) ? slab_s[(c - W) * 64 + lane] : 0.0f;
          if (c <= 1) {  // columns 0 and 1 only exist for c <= 1
            if (i == 1 && j == 1) capM = o[2];
            if (i == 1 && j == 0) capX = pX[c];
            if (i == 0 && j == 1) capY = o[3];
          }
        }
        if (it < WR) {
          const int c = WR - 1 - it;
          bm1prev = o[0];
          by0prev = by0now;
          insprev = insnow;
          pX[c] = o[1];
        }
      }
      dgM = rM;
      firstM = pM[0];
      firstY = yprev;
    }
  }
#if defined(PAIR_EXP_STOP) && PAIR_EXP_STOP == 2  // tuning experiment: forward + backward
  if (capM != 12345.0f) return;
#endif
  // ComputeTotalProbability, :349-364
  capM = __shfl(capM, g * G + (1 / WR));
  capX = __shfl(capX, g * G);
  capY = __shfl(capY, g * G + (1 / WR));
  float totB = fM11 + capM;
  totB = pc_log_add_s(tab, totB, fX10 + capX);
  totB = pc_log_add_s(tab, totB, fY01 + capY);
  const float total = (totF + totB) / 2;

  // sweeps 3 + 4 (pair_sweeps.h): ComputePosteriorMatrix :395 = EXP(min(LOG_ONE, F+B-total)), then sparse outputs
  auto post = [total, tab](const float (&sv)[WR], float (&p)[WR]) {
    float e[WR];
#pragma unroll
    for (int c = 0; c < WR; ++c) {
      const float v = (sv[c] - total) * (1.0f / PC_SCALE);  // back from the 32x domain (exact)
      e[c] = v < 0.0f ? v : 0.0f;
    }
    pc_exp_n<WR>(tab, e, p);
  };
  // sparse outputs through per-lane entry lists (pair_sweeps.h); with th near 0 (dense outputs), or when a list
  // is full, the plane-and-rescan form
  if (a.th < 0.002f || !pair_finish<G, W, WR, false>(a, slab, list, list_cap, s_rowptr, lane, t, g, L1, L2, nsteps, act, task, a.th, post))
    (void)pair_finish<G, W, WR, true>(a, slab, list, list_cap, s_rowptr, lane, t, g, L1, L2, nsteps, act, task, a.th, post);
}

// OCC = wavefronts per SIMD the register allocation must leave room for (the planner reads the result back
// from the code object).  Resident wavefronts are what keeps the vector pipe busy between the lookups of the
// log-sum-exps: 2 per SIMD reach a third of its issue rate.
template <int G, int W, int OCC>
__global__ __launch_bounds__(256, OCC) void k_pairhmm3(dafs_pairhmm3_args a, uint32_t slab_steps, uint32_t rp_cap) {
  constexpr int NG = 64 / G;  // pairs per wavefront
  extern __shared__ uint32_t s_dyn[];  // per (wave, group): rp_cap row pointers
  __shared__ float s_match[56];
  __shared__ float s_ins[8];
  __shared__ float2 s_mi[56];
  __shared__ pc_tables s_tab;
  pc_tables_init(&s_tab, threadIdx.x);
  if (threadIdx.x < 56) s_match[threadIdx.x] = (&a.model.match[0][0])[threadIdx.x];
  if (threadIdx.x < 8) s_ins[threadIdx.x] = a.model.ins[threadIdx.x];
  if (threadIdx.x < 56) s_mi[threadIdx.x] = make_float2((&a.model.match[0][0])[threadIdx.x], a.model.ins[threadIdx.x & 7]);
  __syncthreads();
  pc_tables_init2(&s_tab, threadIdx.x, blockDim.x);
  __syncthreads();

  const int lane = threadIdx.x & 63;
  const int t = lane % G;
  const int g = lane / G;
  const int wave_in_wg = threadIdx.x >> 6;
  // wave-uniform slab base (readfirstlane: the compiler cannot see that threadIdx.x >> 6 is uniform): slab accesses
  // then take a scalar base + the lane's 4*lane + an immediate, with no vector address arithmetic
  const uint32_t wave = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + wave_in_wg);
  // two planes per wave: the DP slab and the entry lists of sweep 3 (pair_sweeps.h)
  const size_t plane = (size_t)slab_steps * W * 64;
  float* __restrict__ slab = a.scratch + (size_t)wave * plane * 2;
  float* __restrict__ list = slab + plane;
  const int list_cap = (int)(slab_steps * W);
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
    // wave-uniform step count: max over the groups of this wave
    int maxL1 = L1;
#pragma unroll
    for (int o = G; o < 64; o <<= 1) maxL1 = max(maxL1, __shfl_xor(maxL1, o));
    const int nsteps = __builtin_amdgcn_readfirstlane(maxL1) + G;
    // columns per lane of this wave: W, or W-1 when its pairs fit (pair_sweeps.h)
    // columns per lane of this wave: W, or W-1 when its pairs fit (pair_sweeps.h)
    if (pair_width<G, W>(L2) == W)
      pairhmm3_pair<G, W, W>(a, &s_tab, s_match, s_ins, s_mi, slab, list, list_cap, s_rowptr, lane, t, g, L1, L2, nsteps, act, task, s1, s2);
    else
      pairhmm3_pair<G, W, (W > 1 ? W - 1 : 1)>(a, &s_tab, s_match, s_ins, s_mi, slab, list, list_cap, s_rowptr, lane, t, g, L1, L2, nsteps, act, task, s1, s2);
  }
}

// instances: (G, W, wavefronts per SIMD the register allocation is held to)
#define V(G, W, OCC) {G, W, (const void*)k_pairhmm3<G, W, OCC>, 0, OCC}
#ifdef PAIR_ONLY_VARIANTS  // tuning builds: -DPAIR_ONLY_VARIANTS="V(32,6,4),V(64,3,4)" compiles in seconds
static pair_variant k_variants[] = {PAIR_ONLY_VARIANTS};
#else
static pair_variant k_variants[] = {
    // four wavefronts per SIMD while the sweeps' registers fit 128 without spilling inside the loops (W <= 6), else three
    // (N = 256, L ~ 200: <32,7> runs 13.8 ms held to 128 registers, 10.7 ms at 168), two from W = 11 on
    V(16, 2, 4), V(16, 3, 4), V(16, 4, 4), V(16, 5, 4), V(16, 6, 4), V(16, 7, 3), V(16, 8, 3), V(16, 10, 2), V(16, 11, 2), V(16, 12, 2), V(16, 14, 1), V(16, 16, 1),
    V(32, 2, 4), V(32, 3, 4), V(32, 4, 4), V(32, 5, 4), V(32, 6, 4), V(32, 7, 3), V(32, 8, 3), V(32, 10, 2), V(32, 12, 2), V(32, 14, 1), V(32, 16, 1),
    V(64, 1, 4), V(64, 2, 4), V(64, 3, 4), V(64, 4, 4), V(64, 5, 4), V(64, 6, 4), V(64, 7, 3), V(64, 8, 3), V(64, 10, 2), V(64, 12, 2), V(64, 14, 1), V(64, 16, 1), V(64, 24, 1), V(64, 32, 1),
};
#endif
#undef V
static const int k_nvariants = (int)(sizeof k_variants / sizeof k_variants[0]);

}  // namespace dafs

using namespace dafs;

extern "C" int dafs_hipk_pairhmm_plan(uint32_t ntasks, uint32_t max_len1, uint32_t max_len2, dafs_pairhmm_plan* plan) {
  if (!plan || ntasks == 0 || max_len1 == 0 || max_len2 == 0) return DAFS_HIP_EINVAL;
  return pair_choose(k_variants, k_nvariants, ntasks, max_len1, max_len2, 2, 600.0, 370.0, 0.83, 0.17, plan);  // two planes: slab + entry lists
}

extern "C" int dafs_hipk_pairhmm3_launch(const dafs_pairhmm3_args* args, const dafs_pairhmm_plan* plan, void* hip_stream) {
  if (!args || !plan) return DAFS_HIP_EINVAL;
  if (args->ntasks == 0) return DAFS_HIP_OK;
  const pair_variant* v = nullptr;
  for (const pair_variant& c : k_variants)
    if (c.G == (int)plan->group && c.W == (int)plan->width) v = &c;
  if (!v || plan->nwaves % 4) return DAFS_HIP_EINVAL;
  const uint32_t rp_cap = plan->slab_steps - plan->group + 1;  // max_len1 + 1 row pointers
  const size_t lds = (size_t)4 * (64 / v->G) * rp_cap * sizeof(uint32_t);
  if (lds > 60 * 1024) return DAFS_HIP_ETOOLONG;
  dafs_pairhmm3_args a = *args;
  // the kernel works on log-probabilities scaled by 32 (pc_math.h, pc_log_add_n): exact, and undone before EXP
  {
    float* f = &a.model.init[0];
    for (size_t k = 0; k < sizeof a.model / sizeof(float); ++k) f[k] *= 32.0f;
  }
  uint32_t steps = plan->slab_steps, cap = rp_cap;
  void* params[] = {&a, &steps, &cap};
  hipError_t launch_err = hipSuccess;
  STAGE_LAUNCH(dafs::ST_PAIRHMM3, (hipStream_t)hip_stream) launch_err = hipLaunchKernel(v->fn, dim3(plan->nwaves / 4), dim3(256), params, lds, (hipStream_t)hip_stream);
  if (hip_check(launch_err)) return DAFS_HIP_ELAUNCH;
  return DAFS_HIP_OK;
}
