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

// ---- table-driven forms: same polynomials, coefficients fetched from LDS by one ds_read_b128
// instead of a dozen selects.  pc_tables lives in LDS, filled by pc_tables_init.
struct pc_tables {
  float4 lookup[12];   // LOOKUP cubic (k3,k2,k1,k0) valid on ((k-1)/2, k/2], k = min(ceil(2x), 10)
  // the same cubics coefficient by coefficient (lk[0] = k3 ... lk[3] = k0): lanes that differ in k hit
  // different banks, lanes that agree broadcast, so the reads are conflict-free.  The rows are 1040 bytes
  // apart on purpose: closer, the compiler fuses the reads of one argument into ds_read2 pairs and then
  // spends moves re-pairing them by coefficient for the packed multiply-adds.
  float lk[4][260];
  float4 lq[16];       // the cubics again, whole, in the slot order of pc_slot16: one ds_read_b128 per LOG_ADD
  double2 exp_hi[8];   // k4,k3 of the EXP quartics, index = clamp(exponent(|x|) + 2, 0, 6); [6] = zero
  double2 exp_mid[8];  // k2,k1
  double exp_lo[8];    // k0
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
      t->exp_lo[k] = z ? 0.0 : e[k][4];
    }
  }
}

// second stage of the initialisation (after a barrier): the coefficient-wise copies, slot (16 - k) & 15 for
// k = ceil(2d) = 0..15 (pc_slot4); k >= 10 is the last piece
__device__ __forceinline__ void pc_tables_init2(pc_tables* t, int tid, int nthreads) {
  for (int k = tid; k < 16; k += nthreads) {
    const float4 a = t->lookup[k < 11 ? k : 11];
    const int e = (16 - k) & 15;
    t->lk[0][e] = a.x; t->lk[1][e] = a.y; t->lk[2][e] = a.z; t->lk[3][e] = a.w;
    t->lq[e] = a;
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

// Two independent LOG_ADDs at once, (x.x (+) y.x, x.y (+) y.y), on the packed FP32 pipe: v_pk_mul_f32 /
// v_pk_add_f32 do both lanes' multiply or add in one issue slot, and with contraction off they stay
// separate roundings, i.e. the same operations as two calls of pc_log_add_t.
typedef float pc_f2 __attribute__((ext_vector_type(2)));
// Slot of the cubic for 2d in the coefficient rows: -ceil(2d) is one instruction (v_cvt_flr_i32_f32 of the negated
// operand; floor(-x) = -ceil(x)), and its low four bits are a valid slot for every d: 0 for d = 0, 16 - ceil(2d)
// for 0 < d < 7.5 (pc_tables_init2 lays the rows out that way), anything in 0..15 for the d >= 7.5 that the
// caller's select throws away (LOG_ZERO operands give 4e20 and saturate the conversion).  Returns the byte offset.
__device__ __forceinline__ unsigned pc_slot4(float d2) {
  int k;
  asm("v_cvt_flr_i32_f32_e64 %0, -%1" : "=v"(k) : "v"(d2));
  return ((unsigned)k << 2) & 60u;
}
__device__ __forceinline__ pc_f2 pc_log_add2_t(const pc_tables* t, pc_f2 x, pc_f2 y) {
  pc_f2 lo, hi;
  lo.x = fminf(x.x, y.x); lo.y = fminf(x.y, y.y);
  hi.x = fmaxf(x.x, y.x); hi.y = fmaxf(x.y, y.y);
  const pc_f2 d = hi - lo;
  const pc_f2 d2 = d + d;
  const unsigned ka = pc_slot4(d2.x), kb = pc_slot4(d2.y);
  const char* b0 = (const char*)t->lk[0], *b1 = (const char*)t->lk[1], *b2 = (const char*)t->lk[2], *b3 = (const char*)t->lk[3];
  const pc_f2 k3 = {*(const float*)(b0 + ka), *(const float*)(b0 + kb)}, k2 = {*(const float*)(b1 + ka), *(const float*)(b1 + kb)};
  const pc_f2 k1 = {*(const float*)(b2 + ka), *(const float*)(b2 + kb)}, k0 = {*(const float*)(b3 + ka), *(const float*)(b3 + kb)};
  const pc_f2 r = (((k3 * d + k2) * d + k1) * d + k0) + lo;
  pc_f2 o;
  o.x = d.x >= 7.5f ? hi.x : r.x;
  o.y = d.y >= 7.5f ? hi.y : r.y;
  return o;
}
// LOG_ADD with one 16-byte lookup and plain (unpacked) float arithmetic.  Measured on MI355X
// (profiles/r02_*_valu_rate.txt): with four or more wavefronts per SIMD v_add_f32 / v_mul_f32 issue at 2.5x the rate of
// every other vector instruction, packed f32 included, and an LDS read costs its SIMD as much as 4-7 vector
// instructions whatever its width -- so: one ds_read_b128 instead of four ds_read_b32, seven plain adds/multiplies,
// and as few other instructions as the reference's semantics allow.
// Slot: (-ceil(32 d)) & 0xF0 = 16 * ((-ceil(2d)) & 15) for d < 7.5 (ceil(ceil(x)/16) = ceil(x/16)), i.e. the byte offset
// of the cubic in lq, from one conversion and one AND; d >= 7.5 lands on some valid slot and is thrown away.
__device__ __forceinline__ float pc_log_add_q(const pc_tables* t, float x, float y) {
  const float lo = fminf(x, y);
  const float hi = fmaxf(x, y);
  const float d = hi - lo;
  const float d32 = d * 32.0f;
  int k;
  asm("v_cvt_flr_i32_f32_e64 %0, -%1" : "=v"(k) : "v"(d32));
  const float4 q = *(const float4*)((const char*)t->lq + (k & 240));
  const float r = ((q.x * d + q.y) * d + q.z) * d + q.w + lo;
  return d >= 7.5f ? hi : r;
}

template <bool LUT>
__device__ __forceinline__ pc_f2 pc_log_add2(const pc_tables* t, pc_f2 x, pc_f2 y) {
  if (LUT) return pc_log_add2_t(t, x, y);
  pc_f2 o;
  o.x = pc_log_add_t<false>(t, x.x, y.x);
  o.y = pc_log_add_t<false>(t, x.y, y.y);
  return o;
}

// EXP (ScoreType.h:37-57) for x <= 0.  Piece boundaries are -1/2, -1, -2, -4, -8, -16, i.e. the
// binary exponent of |x| selects the piece; index 6 holds zeros (x <= -16 returns 0).
__device__ __forceinline__ float pc_exp_t(const pc_tables* t, float xf) {
  const float a = fmaxf(-xf, 1.0e-30f);                   // |x|, away from 0 so that the exponent is defined
  int e;
  (void)frexpf(a, &e);                                     // a = m * 2^e, m in [0.5, 1)
  // x > -0.5 <=> a < 0.5 <=> e <= -1 ; -1 < x <= -0.5 <=> e == 0 ; ... ; -16 < x <= -8 <=> e == 4
  // (piece boundaries are closed on the lower side: x <= -0.5 leaves piece 0, matching a >= 0.5 <=> e >= 0)
  const int piece = min(max(e + 1, 0), 6);
  const double2 h = t->exp_hi[piece], m = t->exp_mid[piece];
  const double c = t->exp_lo[piece];
  const double x = (double)xf;
  return (float)((((h.x * x + h.y) * x + m.x) * x + m.y) * x + c);
}

}  // namespace dafs
