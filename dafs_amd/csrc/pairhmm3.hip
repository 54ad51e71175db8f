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
#include "pair_sweeps.h"

namespace dafs {

// OCC = wavefronts per SIMD the register allocation must leave room for (the planner reads the result back
// from the code object).  The sweeps wait on an LDS lookup in every log-sum-exp, so resident wavefronts are
// what keeps the vector pipe busy: 2 per SIMD reach a third of its issue rate.
template <int G, int W, int OCC>
__global__ __launch_bounds__(256, OCC) void k_pairhmm3(dafs_pairhmm3_args a, uint32_t slab_steps, uint32_t rp_cap) {
  constexpr bool LUT = true;
  constexpr int NG = 64 / G;  // pairs per wavefront
  extern __shared__ uint32_t s_dyn[];  // per (wave, group): rp_cap row pointers
  __shared__ float s_match[56];
  __shared__ float s_ins[8];
  __shared__ pc_tables s_tab;
  pc_tables_init(&s_tab, threadIdx.x);
  if (threadIdx.x < 56) s_match[threadIdx.x] = (&a.model.match[0][0])[threadIdx.x];
  if (threadIdx.x < 8) s_ins[threadIdx.x] = a.model.ins[threadIdx.x];
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
  float* __restrict__ slab = a.scratch + (size_t)wave * slab_steps * W * 64;
  uint32_t* __restrict__ s_rowptr = s_dyn + (size_t)(wave_in_wg * NG + g) * rp_cap;

  const float LZ = PC_LOG_ZERO;
  const float i0 = a.model.init[0], i1 = a.model.init[1], i2 = a.model.init[2];
  const float tMM = a.model.trans[0][0], tMX = a.model.trans[0][1], tMY = a.model.trans[0][2];
  const float tXM = a.model.trans[1][0], tXX = a.model.trans[1][1];
  const float tYM = a.model.trans[2][0], tYY = a.model.trans[2][2];
  const float th = a.th;

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
    const int wr = pair_width<G, W>(L2);
    const bool full = wr == W;
    const int j0 = t * wr;

    // residue classes of this lane's columns: cc[c] = class of column j=t*W+c (s2[j-1]); 6 = other
    int cc[W + 1];
#pragma unroll
    for (int c = 0; c <= W; ++c) {
      const int j = j0 + c;
      cc[c] = (j >= 1 && j <= L2) ? (int)s2[j - 1] : 6;
    }
    const int tlast = (L2 >= 0 ? L2 : 0) / wr;  // lane (within group) that owns column L2
    float ins2[W + 1];  // single-emission score of this lane's columns (a register instead of an LDS lookup per cell)
#pragma unroll
    for (int c = 0; c <= W; ++c) ins2[c] = s_ins[cc[c]];
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
    float endM = LZ, endX = LZ, endY = LZ;  // F_k(L1, L2)
    {
      float pM[W], pX[W], pY[W];
#pragma unroll
      for (int c = 0; c < W; ++c) pM[c] = pX[c] = pY[c] = LZ;
      float lastM = LZ, lastX = LZ, lastY = LZ;  // this lane's last column, row of the previous step
      float dgM = LZ, dgX = LZ, dgY = LZ;        // neighbour's last column, one row earlier
      for (int s = 0; s < nsteps; ++s) {
        const int i = s - t;
        const bool rowv = (i >= 0) && (i <= L1);
        float* __restrict__ slab_s = slab + (size_t)s * (W * 64);
        const int c1 = (rowv && i >= 1) ? (int)s1[i - 1] : 6;
        const float rM = shift_up1<G>(lastM, LZ, t), rX = shift_up1<G>(lastX, LZ, t), rY = shift_up1<G>(lastY, LZ, t);
        float dM = dgM, dX = dgX, dY = dgY;  // (i-1, j-1)
        float lM = rM, lY = rY;              // (i, j-1)
        const float insc1 = s_ins[c1];
#pragma unroll
        for (int c = 0; c < W; ++c) {
          if (c < W - 1 || full) {
            const int j = j0 + c;
            const float mt = s_match[c1 * 8 + cc[c]];
            // ProbabilisticModel.h:152-155 (M), :159-161 (X), :165-167 (Y): four LOG_ADDs
            const float m1 = pc_log_add_q(&s_tab, dM + tMM, dX + tXM);
            const float xx = pc_log_add_q(&s_tab, pM[c] + tMX, pX[c] + tXX);
            const float m2 = pc_log_add_q(&s_tab, m1, dY + tYM);
            const float yy = pc_log_add_q(&s_tab, lM + tMY, lY + tYY);
            float m = m2 + mt;
            float x = insc1 + xx;
            float y = ins2[c] + yy;
            if (c <= 1) {  // j <= 1 needs c <= 1: initial cells, :123-131 (cells with i<=1 && j<=1 are skipped by :150)
              if (i <= 1 && j <= 1) {
                m = (i == 1 && j == 1) ? fM11 : LZ;
                x = (i == 1 && j == 0) ? fX10 : LZ;
                y = (i == 0 && j == 1) ? fY01 : LZ;
              }
            }
            dM = pM[c]; dX = pX[c]; dY = pY[c];
            pM[c] = m; pX[c] = x; pY[c] = y;
            lM = m; lY = y;
            slab_s[c * 64 + lane] = m;  // slot (s,c,lane) is private to this lane: no guard needed
          }
        }
        dgM = rM; dgX = rX; dgY = rY;
        lastM = lM; lastX = full ? pX[W - 1] : pX[W > 1 ? W - 2 : 0]; lastY = lY;
        if (i == L1 && t == tlast) {  // F_k(L1, L2): one lane of the group, once (a rare branch instead of three selects per cell)
          const int cl = L2 - j0;
#pragma unroll
          for (int c = 0; c < W; ++c)
            if (c == cl) { endM = pM[c]; endX = pX[c]; endY = pY[c]; }
        }
      }
    }
#if defined(PAIR_EXP_STOP) && PAIR_EXP_STOP == 1  // tuning experiment: time of the forward sweep alone
    if (endM != 12345.0f) continue;
#endif
    float totF = LZ;  // ComputeTotalProbability, :341-347 (B_k(L1,L2) = init_k)
    totF = pc_log_add_t<LUT>(&s_tab, totF, endM + i0);
    totF = pc_log_add_t<LUT>(&s_tab, totF, endX + i1);
    totF = pc_log_add_t<LUT>(&s_tab, totF, endY + i2);
    totF = __shfl(totF, g * G + tlast);

    // ------------------------------------------------------------------ sweep 2: backward
    float capM = LZ, capX = LZ, capY = LZ;  // B_M(1,1), B_X(1,0), B_Y(0,1)
    {
      float pM[W], pX[W];
#pragma unroll
      for (int c = 0; c < W; ++c) pM[c] = pX[c] = LZ;
      float firstM = LZ, firstY = LZ;  // this lane's first column, row of the previous step
      float dgM = LZ;                  // right neighbour's first column, one row later
      for (int s = 0; s < nsteps; ++s) {
        // mirrored skew: lane G-1 starts with the last row of the longest pair of the wave (rows beyond a shorter
        // pair's L1 stay LOG_ZERO like every other cell outside the grid), so that the forward step that
        // stored row i for this lane, sf = i + t, is the same for the whole wave
        const int sf = nsteps - 1 - s;
        const int i = sf - t;
        const bool rowv = (i >= 0) && (i <= L1);
        float* __restrict__ slab_s = slab + (size_t)sf * (W * 64);
        const int c1 = (rowv && i < L1) ? (int)s1[i] : 6;
        const float rM = shift_down1<G>(firstM, LZ, t), rY = shift_down1<G>(firstY, LZ, t);
        float dM = dgM;  // B_M(i+1, j+1)
        float rgY = rY;  // B_Y(i, j+1)
        const float insc1 = s_ins[c1];
        float fwd[W];
#pragma unroll
        for (int c = 0; c < W; ++c) {
          const int j = j0 + c;
          if (c < W - 1 || full) fwd[c] = (rowv && j <= L2) ? slab_s[c * 64 + lane] : 0.0f;
        }
#pragma unroll
        for (int c = W - 1; c >= 0; --c) {
          if (!(c < W - 1 || full)) continue;
          const int j = j0 + c;
          const int c2 = cc[c + 1];  // class of s2[j] (iter2[j+1]); 'other' beyond the end
          // :233-237.  The cells start at LOG_ZERO (:213-214 sets the corner to the initial distribution), and
          // LOG_ZERO (+) v == v for every v >= LOG_ZERO (the d >= 7.5 exit; v == LOG_ZERO gives LOG_ZERO), so
          // the first accumulation is a plain assignment; at the corner the right-hand sides are LOG_ZERO
          // (nothing lies beyond it) and the accumulation would return the initial values unchanged.
          const float pxy = dM + s_match[c1 * 8 + c2];
          float bm = pxy + tMM, bx = pxy + tXM, by = pxy + tYM;
          if (i == L1 && j == L2) { bm = i0; bx = i1; by = i2; }
          // :238-243 (M and X take the X-step term) and :244-249 (M and Y take the Y-step term)
          const float tx = pX[c] + insc1;
          bm = pc_log_add_q(&s_tab, bm, tx + tMX);
          bx = pc_log_add_q(&s_tab, bx, tx + tXX);
          const float ty = rgY + ins2[c + 1];
          bm = pc_log_add_q(&s_tab, bm, ty + tMY);
          by = pc_log_add_q(&s_tab, by, ty + tYY);
          dM = pM[c];
          pM[c] = bm; pX[c] = bx;
          rgY = by;
          if (rowv && j <= L2) slab_s[c * 64 + lane] = fwd[c] + bm;  // forward[ij] + backward[ij], :395
          if (c <= 1) {  // columns 0 and 1 only exist for c <= 1
            if (i == 1 && j == 1) capM = bm;
            if (i == 1 && j == 0) capX = bx;
            if (i == 0 && j == 1) capY = by;
          }
          if (c == 0) firstY = by;
        }
        dgM = rM;
        firstM = pM[0];
      }
    }
#if defined(PAIR_EXP_STOP) && PAIR_EXP_STOP == 2  // tuning experiment: forward + backward
    if (capM != 12345.0f) continue;
#endif
    // ComputeTotalProbability, :349-364
    capM = __shfl(capM, g * G + (1 / wr));
    capX = __shfl(capX, g * G);
    capY = __shfl(capY, g * G + (1 / wr));
    float totB = fM11 + capM;
    totB = pc_log_add_t<LUT>(&s_tab, totB, fX10 + capX);
    totB = pc_log_add_t<LUT>(&s_tab, totB, fY01 + capY);
    const float total = (totF + totB) / 2;

    // sweeps 3 + 4 (pair_sweeps.h): ComputePosteriorMatrix :395 = EXP(min(LOG_ONE, F+B-total)), then sparse outputs
    const pc_tables* tab = &s_tab;
    pair_finish<G, W>(a, slab, s_rowptr, lane, t, g, L1, L2, nsteps, wr, act, task, th, [total, tab](float sv) {
      const float e = sv - total;
      return pc_exp_t(tab, e < 0.0f ? e : 0.0f);
    });
  }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
// instances: (G, W, wavefronts per SIMD the register allocation is held to)
#define V(G, W, OCC) {G, W, (const void*)k_pairhmm3<G, W, OCC>, 0}
static pair_variant k_variants[] = {
    V(16, 2, 4), V(16, 3, 4), V(16, 4, 4), V(16, 5, 4), V(16, 6, 3), V(16, 7, 3), V(16, 8, 3), V(16, 10, 3), V(16, 11, 2), V(16, 12, 2), V(16, 14, 2), V(16, 16, 2),
    V(32, 2, 4), V(32, 3, 4), V(32, 4, 4), V(32, 5, 4), V(32, 6, 4), V(32, 7, 4), V(32, 8, 3), V(32, 10, 3), V(32, 12, 2), V(32, 14, 2), V(32, 16, 2),
    V(64, 1, 4), V(64, 2, 4), V(64, 3, 4), V(64, 4, 4), V(64, 5, 4), V(64, 6, 4), V(64, 7, 4), V(64, 8, 3), V(64, 10, 3), V(64, 12, 3), V(64, 14, 2), V(64, 16, 2), V(64, 24, 1), V(64, 32, 1),
};
#undef V
static const int k_nvariants = (int)(sizeof k_variants / sizeof k_variants[0]);

}  // namespace dafs

using namespace dafs;

extern "C" int dafs_hipk_pairhmm_plan(uint32_t ntasks, uint32_t max_len1, uint32_t max_len2, dafs_pairhmm_plan* plan) {
  if (!plan || ntasks == 0 || max_len1 == 0 || max_len2 == 0) return DAFS_HIP_EINVAL;
  // instruction counts of the four sweeps together, from the ISA of the W = 5 / 6 instances
  return pair_choose(k_variants, k_nvariants, ntasks, max_len1, max_len2, 1, 150.0, 240.0, plan);
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
  uint32_t steps = plan->slab_steps, cap = rp_cap;
  void* params[] = {&a, &steps, &cap};
  if (hip_check(hipLaunchKernel(v->fn, dim3(plan->nwaves / 4), dim3(256), params, lds, (hipStream_t)hip_stream))) return DAFS_HIP_ELAUNCH;
  return DAFS_HIP_OK;
}
