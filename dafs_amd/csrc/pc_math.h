// dafs_amd/csrc/pc_math.h -- ProbCons log-space arithmetic for device code.
//
// The approximations the reference pair-HMM uses (reference src/probconsRNA/ScoreType.h):
// every literal keeps the reference's type (float-suffixed in LOOKUP, double in EXP) and every
// expression keeps its association; the translation unit is compiled with -ffp-contract=off so
// no multiply-add is fused (the reference build is x86-64 baseline SSE2, no FMA).
#pragma once
#include <hip/hip_runtime.h>

namespace dafs {

#define PC_LOG_ZERO (-2e20f)  // ScoreType.h:18

// log(exp(x)+1) for 0 <= x <= 7.5: four float cubics (ScoreType.h:187-198)
__device__ __forceinline__ float pc_lookup(float x) {
  const bool a = x <= 1.00f, b = x <= 2.50f, c = x <= 4.50f;
  const float k3 = a ? -0.009350833524763f : b ? -0.014532321752540f : c ? -0.004605031767994f : -0.000458661602210f;
  const float k2 = a ? 0.130659527668286f : b ? 0.139942324101744f : c ? 0.063427417320019f : 0.009695946122598f;
  const float k1 = a ? 0.498799810682272f : b ? 0.495635523139337f : c ? 0.695956496475118f : 0.930734667215156f;
  const float k0 = a ? 0.693203116424741f : b ? 0.692140569840976f : c ? 0.514272634594009f : 0.168037164329057f;
  return ((k3 * x + k2) * x + k1) * x + k0;
}

// LOG_ADD (ScoreType.h:259-262); LOG_PLUS_EQUALS (:233-238) is x = pc_log_add(x, y):
// both branch on x < y and apply the same two exits.
__device__ __forceinline__ float pc_log_add(float x, float y) {
  const bool lt = x < y;
  const float lo = lt ? x : y;  // the smaller (x when x<y, else y)
  const float hi = lt ? y : x;
  const float d = hi - lo;
  const float r = pc_lookup(d) + lo;
  return (lo == PC_LOG_ZERO || d >= 7.5f) ? hi : r;
}

// EXP (ScoreType.h:37-57): quartic pieces evaluated in double, narrowed to float.  x <= 0 here.
__device__ __forceinline__ float pc_exp(float xf) {
  const double x = (double)xf;
  double k4, k3, k2, k1, k0;
  if (xf > -2) {
    if (xf > -0.5) {
      k4 = 0.03254409303190190000; k3 = 0.16280432765779600000; k2 = 0.49929760485974900000; k1 = 0.99995149601363700000; k0 = 0.99999925508501600000;
    } else if (xf > -1) {
      k4 = 0.01973899026052090000; k3 = 0.13822379685007000000; k2 = 0.48056651562365000000; k1 = 0.99326940370383500000; k0 = 0.99906756856399500000;
    } else {
      k4 = 0.00940528203591384000; k3 = 0.09414963667859410000; k2 = 0.40825793595877300000; k1 = 0.93933625499130400000; k0 = 0.98369508190545300000;
    }
  } else if (xf > -8) {
    if (xf > -4) {
      k4 = 0.00217245711583303000; k3 = 0.03484829428350620000; k2 = 0.22118199801337800000; k1 = 0.67049462206469500000; k0 = 0.83556950223398500000;
    } else {
      k4 = 0.00012398771025456900; k3 = 0.00349155785951272000; k2 = 0.03727721426017900000; k1 = 0.17974997741536900000; k0 = 0.33249299994217400000;
    }
  } else if (xf > -16) {
    k4 = 0.00000051741713416603; k3 = 0.00002721456879608080; k2 = 0.00053418601865636800; k1 = 0.00464101989351936000; k0 = 0.01507447981459420000;
  } else {
    return 0.0f;
  }
  return (float)((((k4 * x + k3) * x + k2) * x + k1) * x + k0);
}

// ---- table-driven forms: same polynomials, coefficients fetched from LDS.  pc_tables lives in LDS, filled by
// pc_tables_init (thread 0) + pc_tables_init2 (after a barrier).
struct pc_tables {
  float4 lookup[12];   // LOOKUP cubic (k3,k2,k1,k0) valid on ((k-1)/2, k/2], k = min(ceil(2x), 10)
  float4 lq[16];       // the same cubics in the slot order of pc_log_add_n, for arguments scaled by 32: one ds_read_b128 per LOG_ADD
  double2 exp_hi[8];   // k4,k3 of the EXP quartics, index = clamp(exponent(|x|) + 2, 0, 6); [6] = zero
  double2 exp_mid[8];  // k2,k1
  double2 exp_lo[8];   // k0 (and a pad: the three tables share one byte offset, 16 * piece)
};

__device__ __forceinline__ void pc_tables_init(pc_tables* t, int tid) {
  if (tid == 0) {
    const float4 c0 = make_float4(-0.009350833524763f, 0.130659527668286f, 0.498799810682272f, 0.693203116424741f);  // x <= 1
    const float4 c1 = make_float4(-0.014532321752540f, 0.139942324101744f, 0.495635523139337f, 0.692140569840976f);  // x <= 2.5
    const float4 c2 = make_float4(-0.004605031767994f, 0.063427417320019f, 0.695956496475118f, 0.514272634594009f);  // x <= 4.5
    const float4 c3 = make_float4(-0.000458661602210f, 0.009695946122598f, 0.930734667215156f, 0.168037164329057f);  // else
    for (int k = 0; k <= 10; ++k) t->lookup[k] = k <= 2 ? c0 : (k <= 5 ? c1 : (k <= 9 ? c2 : c3));
    t->lookup[11] = c3;
    const double e[6][5] = {
        {0.03254409303190190000, 0.16280432765779600000, 0.49929760485974900000, 0.99995149601363700000, 0.99999925508501600000},
        {0.01973899026052090000, 0.13822379685007000000, 0.48056651562365000000, 0.99326940370383500000, 0.99906756856399500000},
        {0.00940528203591384000, 0.09414963667859410000, 0.40825793595877300000, 0.93933625499130400000, 0.98369508190545300000},
        {0.00217245711583303000, 0.03484829428350620000, 0.22118199801337800000, 0.67049462206469500000, 0.83556950223398500000},
        {0.00012398771025456900, 0.00349155785951272000, 0.03727721426017900000, 0.17974997741536900000, 0.33249299994217400000},
        {0.00000051741713416603, 0.00002721456879608080, 0.00053418601865636800, 0.00464101989351936000, 0.01507447981459420000}};
    for (int k = 0; k < 8; ++k) {
      const bool z = k >= 6;
      t->exp_hi[k] = z ? make_double2(0.0, 0.0) : make_double2(e[k][0], e[k][1]);
      t->exp_mid[k] = z ? make_double2(0.0, 0.0) : make_double2(e[k][2], e[k][3]);
      t->exp_lo[k] = make_double2(z ? 0.0 : e[k][4], 0.0);
    }
  }
}

// second stage of the initialisation (after a barrier): slot (16 - k) & 15 holds the cubic of k = ceil(2d) = 0..15
// (k >= 10 is the last piece), rescaled for the 32x domain of pc_log_add_n (powers of two: exact)
__device__ __forceinline__ void pc_tables_init2(pc_tables* t, int tid, int nthreads) {
  for (int k = tid; k < 16; k += nthreads) {
    const float4 a = t->lookup[k < 11 ? k : 11];
    t->lq[(16 - k) & 15] = make_float4(a.x * (1.0f / 1024.0f), a.y * (1.0f / 32.0f), a.z, a.w * 32.0f);
  }
}

// LOOKUP (ScoreType.h:187-198).  The breakpoints 1.0, 2.5, 4.5 are multiples of 1/2 and every piece
// is closed on the right, so the piece is a function of ceil(2x): no compare chain.
__device__ __forceinline__ float pc_lookup_t(const pc_tables* t, float x) {
  const unsigned k = (unsigned)fminf(ceilf(x * 2.0f), 10.0f);  // x >= 0 here
  const float4 q = t->lookup[k];
  return ((q.x * x + q.y) * x + q.z) * x + q.w;
}

// LOG_ADD / LOG_PLUS_EQUALS (ScoreType.h:233-262).  LUT = true fetches the cubic from LDS, LUT = false
// selects it (pc_lookup).  Two identities keep the instruction count down without changing a bit:
//  * min/max instead of compare+selects (no NaN reaches a DP cell);
//  * the reference's `lo == LOG_ZERO` exit is implied by its `hi - lo >= 7.5` exit: LOG_ZERO is
//    -2e20, whose ulp is 1.8e13, so hi - LOG_ZERO >= 7.5 unless hi == LOG_ZERO too, and then both
//    exits give LOG_ZERO (LOOKUP(0) + LOG_ZERO rounds to LOG_ZERO).
template <bool LUT>
__device__ __forceinline__ float pc_log_add_t(const pc_tables* t, float x, float y) {
  const float lo = fminf(x, y);
  const float hi = fmaxf(x, y);
  const float d = hi - lo;
  const float r = (LUT ? pc_lookup_t(t, d) : pc_lookup(d)) + lo;
  return d >= 7.5f ? hi : r;
}

// The form the DP sweeps use: N independent LOG_ADDs at once, o[n] = x[n] (+) y[n], on values SCALED BY 32.
// What the measurements on MI355X say (profiles/r02_a_valu_rate.txt, tools/valu_rate.hip): at two or more wavefronts
// per SIMD the sweeps are bound by instruction issue -- plain f32/int adds, multiplies, ANDs and moves cost ~1.1 ns of
// their SIMD each, every other vector instruction (min/max, compares, selects, conversions, shifts, packed f32, f64)
// ~1.9 ns, an LDS read of up to 16 bytes per lane ~6.9 ns when all four SIMDs of the CU issue them.  Hence:
//  * the 32x domain: every log-probability the sweeps carry is 32 times the reference's value.  Multiplying by a
//    power of two commutes with IEEE rounding (no value here comes near the subnormal or overflow range; LOG_ZERO
//    becomes -6.4e21 and keeps absorbing), so every sum, difference, comparison and the cubic
//    ((k3/1024 * D + k2/32) * D + k1) * D + 32 k0, D = 32 d, are the reference's values times 32, bit for bit;
//    the posterior step divides by 32 again.  What it buys: 32 d = hi - lo needs no multiply, and
//  * one ds_read_b128 per LOG_ADD: slot (-ceil(32 d)) & 0xF0 = 16 * ((-ceil(2d)) & 15) for d < 7.5
//    (ceil(ceil(x)/16) = ceil(x/16)) is the byte offset of the cubic in lq, from one conversion
//    (v_cvt_flr_i32_f32 of the negated operand: floor(-x) = -ceil(x)) and one AND; d >= 7.5 (LOG_ZERO operands give
//    1e22, which saturates the conversion) lands on some valid slot and is thrown away by the select;
//  * seven plain adds/multiplies for the cubic (contraction is off: separate roundings, as in the reference);
//  * the N lookups are issued together, between scheduling barriers -- left alone, the compiler sinks every
//    lookup next to its polynomial and waits N times.
#define PC_SCALE 32.0f
#define PC_LOG_ZERO_S (PC_LOG_ZERO * PC_SCALE)
template <int N>
__device__ __forceinline__ void pc_log_add_n(const pc_tables* t, const float (&x)[N], const float (&y)[N], float (&o)[N]) {
  float lo[N], hi[N], d[N];
  int k[N];
  float4 q[N];
#pragma unroll
  for (int n = 0; n < N; ++n) {
    lo[n] = fminf(x[n], y[n]);
    hi[n] = fmaxf(x[n], y[n]);
    d[n] = hi[n] - lo[n];
    asm("v_cvt_flr_i32_f32_e64 %0, -%1" : "=v"(k[n]) : "v"(d[n]));
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int n = 0; n < N; ++n) {
#if defined(PAIR_EXP_NOLDS)  // tuning experiment: what the lookups cost
    q[n] = make_float4(__int_as_float(k[n]), 0.1f, 0.5f, 0.7f);
#else
    q[n] = *(const float4*)((const char*)t->lq + (k[n] & 240));
#endif
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int n = 0; n < N; ++n) {
#if defined(PAIR_EXP_NOPOLY)  // tuning experiment: what the cubic costs
    const float r = q[n].x + lo[n];
#else
    const float r = ((q[n].x * d[n] + q[n].y) * d[n] + q[n].z) * d[n] + q[n].w + lo[n];
#endif
    o[n] = d[n] >= 7.5f * PC_SCALE ? hi[n] : r;
  }
}
__device__ __forceinline__ float pc_log_add_s(const pc_tables* t, float x, float y) {  // one LOG_ADD in the 32x domain
  const float xs[1] = {x}, ys[1] = {y};
  float o[1];
  pc_log_add_n<1>(t, xs, ys, o);
  return o[0];
}

// EXP (ScoreType.h:37-57) for x <= 0, N values of an array at once (the table reads issued together).  Piece boundaries are
// -1/2, -1, -2, -4, -8, -16, i.e. the binary exponent of |x| selects the piece; index 6 holds zeros (x <= -16
// returns 0).  The quartic is evaluated in double and narrowed, like the reference.
template <int N, int OFF = 0, int TOT = N>
__device__ __forceinline__ void pc_exp_chunk(const pc_tables* t, const float (&xf)[TOT], float (&o)[TOT]) {
  int piece[N];
  double2 h[N], m[N];
  double c[N];
#pragma unroll
  for (int n = 0; n < N; ++n) {
    // |x| = m * 2^e, m in [0.5, 1):  x > -0.5 <=> e <= -1 ; -1 < x <= -0.5 <=> e == 0 ; ... ; -16 < x <= -8 <=> e == 4
    // (piece boundaries are closed on the lower side: x <= -0.5 leaves piece 0, matching |x| >= 0.5 <=> e >= 0), so the
    // piece is clamp(e + 1, 0, 6), and e + 1 is the biased exponent field minus 125 (zero and denormals: field 0 -> piece 0).
    // One bit-field extract, one add, one median instead of frexp and its guards.
    const int e8 = (__float_as_int(xf[OFF + n]) >> 23) & 0xFF;
    piece[n] = min(max(e8 - 125, 0), 6);
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int n = 0; n < N; ++n) { h[n] = t->exp_hi[piece[n]]; m[n] = t->exp_mid[piece[n]]; c[n] = t->exp_lo[piece[n]].x; }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int n = 0; n < N; ++n) {
    const double x = (double)xf[OFF + n];
    o[OFF + n] = (float)((((h[n].x * x + h[n].y) * x + m[n].x) * x + m[n].y) * x + c[n]);
  }
}
// N values, three at a time (a chunk holds ten registers of coefficients per value)
template <int N>
__device__ __forceinline__ void pc_exp_n(const pc_tables* t, const float (&xf)[N], float (&o)[N]) {
  if constexpr (N >= 1) pc_exp_chunk<(N >= 3 ? 3 : N), 0, N>(t, xf, o);
  if constexpr (N >= 4) pc_exp_chunk<(N >= 6 ? 3 : N - 3), 3, N>(t, xf, o);
  if constexpr (N >= 7) pc_exp_chunk<(N >= 9 ? 3 : N - 6), 6, N>(t, xf, o);
  if constexpr (N >= 10) pc_exp_chunk<(N >= 12 ? 3 : N - 9), 9, N>(t, xf, o);
  if constexpr (N >= 13) pc_exp_chunk<(N >= 15 ? 3 : N - 12), 12, N>(t, xf, o);
  if constexpr (N >= 16) pc_exp_chunk<(N >= 18 ? 3 : N - 15), 15, N>(t, xf, o);
  if constexpr (N >= 19) pc_exp_chunk<(N >= 21 ? 3 : N - 18), 18, N>(t, xf, o);
  if constexpr (N >= 22) pc_exp_chunk<(N >= 24 ? 3 : N - 21), 21, N>(t, xf, o);
  if constexpr (N >= 25) pc_exp_chunk<(N >= 27 ? 3 : N - 24), 24, N>(t, xf, o);
  if constexpr (N >= 28) pc_exp_chunk<(N >= 30 ? 3 : N - 27), 27, N>(t, xf, o);
  if constexpr (N >= 31) pc_exp_chunk<(N >= 33 ? 3 : N - 30), 30, N>(t, xf, o);
  static_assert(N <= 33, "pc_exp_n: widen the chunk list");
}

}  // namespace dafs
