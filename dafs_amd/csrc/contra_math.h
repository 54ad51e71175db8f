// dafs_amd/csrc/contra_math.h -- log-space arithmetic shared by the CONTRAfold and CONTRAlign
// kernels: the float instantiations of reference src/contrafold/LogSpace.hpp (the contralign copy
// is identical apart from the namespace).  Every constant is the reference's double literal
// narrowed to float, as float(...) does there; compile with -ffp-contract=off.
#pragma once
#include <hip/hip_runtime.h>

namespace dafs {

#define CONTRA_NEG_INF (-2e20f)  // LogSpace.hpp:13

// LogSpace.hpp:28-60
__device__ __forceinline__ float contra_exp(float x) {
  if (x < (float)(-2.4915033807)) {
    if (x < (float)(-5.8622823336)) {
      if (x < (float)(-9.91152)) return 0.0f;
      return (((float)(0.0000803850) * x + (float)(0.0021627428)) * x + (float)(0.0194708555)) * x + (float)(0.0588080014);
    }
    if (x < (float)(-3.8396630909))
      return (((float)(0.0013889414) * x + (float)(0.0244676474)) * x + (float)(0.1471290604)) * x + (float)(0.3042757740);
    return (((float)(0.0072335607) * x + (float)(0.0906002677)) * x + (float)(0.3983111356)) * x + (float)(0.6245959221);
  }
  if (x < (float)(-0.6725053211)) {
    if (x < (float)(-1.4805375919))
      return (((float)(0.0232410351) * x + (float)(0.2085645908)) * x + (float)(0.6906367911)) * x + (float)(0.8682322329);
    return (((float)(0.0573782771) * x + (float)(0.3580258429)) * x + (float)(0.9121133217)) * x + (float)(0.9793091728);
  }
  if (x < 0.0f)
    return (((float)(0.1199175927) * x + (float)(0.4815668234)) * x + (float)(0.9975991939)) * x + (float)(0.9999505077);
  // x >= 0: the reference calls expf.  Every caller clips the accumulated value to [0,1]
  // (posteriors), and a term exp(x >= 0) >= 1 already saturates the clip, so the last-bit
  // behaviour of the device expf cannot change a result.
  return x > (float)(46.052) ? (float)(1e20) : expf(x);
}

// LogSpace.hpp:74-107: log(exp(x)+1), 0 <= x <= 11.8624794162, eight float cubics
__device__ __forceinline__ float contra_log_exp_plus_one(float x) {
  float k3, k2, k1, k0;
  if (x < (float)(3.3792499610)) {
    if (x < (float)(1.6320158198)) {
      if (x < (float)(0.6615367791)) { k3 = (float)(-0.0065591595); k2 = (float)(0.1276442762); k1 = (float)(0.4996554598); k0 = (float)(0.6931542306); }
      else { k3 = (float)(-0.0155157557); k2 = (float)(0.1446775699); k1 = (float)(0.4882939746); k0 = (float)(0.6958092989); }
    } else if (x < (float)(2.4912588184)) { k3 = (float)(-0.0128909247); k2 = (float)(0.1301028251); k1 = (float)(0.5150398748); k0 = (float)(0.6795585882); }
    else { k3 = (float)(-0.0072142647); k2 = (float)(0.0877540853); k1 = (float)(0.6208708362); k0 = (float)(0.5909675829); }
  } else if (x < (float)(5.7890710412)) {
    if (x < (float)(4.4261691294)) { k3 = (float)(-0.0031455354); k2 = (float)(0.0467229449); k1 = (float)(0.7592532310); k0 = (float)(0.4348794399); }
    else { k3 = (float)(-0.0010110698); k2 = (float)(0.0185943421); k1 = (float)(0.8831730747); k0 = (float)(0.2523695427); }
  } else if (x < (float)(7.8162726752)) { k3 = (float)(-0.0001962780); k2 = (float)(0.0046084408); k1 = (float)(0.9634431978); k0 = (float)(0.0983148903); }
  else { k3 = (float)(-0.0000113994); k2 = (float)(0.0003734731); k1 = (float)(0.9959107193); k0 = (float)(0.0149855051); }
  return ((k3 * x + k2) * x + k1) * x + k0;
}

// Fast_LogPlusEquals, LogSpace.hpp:239-244: returns the new x
__device__ __forceinline__ float contra_lpe(float x, float y) {
  const bool lt = x < y;
  const float hi = lt ? y : x;
  const float lo = lt ? x : y;
  const float d = hi - lo;
  const float r = contra_log_exp_plus_one(d) + lo;
  return (lo > (float)(-2e20 / 2) && d < (float)(11.8624794162)) ? r : hi;
}

// ---- table-driven forms for the throughput-bound pair kernel (same polynomials, same piece limits).
// The pieces have irregular limits, but no two limits are closer than 0.8, so a grid of step 1/2 has at
// most one limit per cell: cell g = floor(2x) stores the number of limits at or below its start and the
// limit inside it (or +inf); piece = base + (x >= limit).  Coefficients sit coefficient by coefficient
// in rows 1040 bytes apart (conflict-free, and not fused into ds_read2 pairs; see pc_math.h).
struct contra_tables {
  float2 lcell[24];     // log(exp(x)+1): {limit inside the cell, pieces below as float bits}
  float lk[4][260];     // k3, k2, k1, k0 of the eight cubics
  float2 ecell[20];     // exp(x), -10 <= x < 0: cell g = floor(2 (x + 10))
  float ek[4][260];     // seven cubics; piece 0 (x < -9.91152) is all zeros
  // one-lookup form for latency-bound chains (contra_lpe_t1): per cell the cubic below the limit, the cubic from the
  // limit on, and the limit (+inf if the cell has none)
  float4 lone[24][3];
};

__device__ __forceinline__ void contra_tables_init(contra_tables* t, int tid) {
  if (tid != 0) return;
  const float lim[7] = {(float)(0.6615367791), (float)(1.6320158198), (float)(2.4912588184), (float)(3.3792499610),
                        (float)(4.4261691294), (float)(5.7890710412), (float)(7.8162726752)};
  const float k[8][4] = {
      {(float)(-0.0065591595), (float)(0.1276442762), (float)(0.4996554598), (float)(0.6931542306)},
      {(float)(-0.0155157557), (float)(0.1446775699), (float)(0.4882939746), (float)(0.6958092989)},
      {(float)(-0.0128909247), (float)(0.1301028251), (float)(0.5150398748), (float)(0.6795585882)},
      {(float)(-0.0072142647), (float)(0.0877540853), (float)(0.6208708362), (float)(0.5909675829)},
      {(float)(-0.0031455354), (float)(0.0467229449), (float)(0.7592532310), (float)(0.4348794399)},
      {(float)(-0.0010110698), (float)(0.0185943421), (float)(0.8831730747), (float)(0.2523695427)},
      {(float)(-0.0001962780), (float)(0.0046084408), (float)(0.9634431978), (float)(0.0983148903)},
      {(float)(-0.0000113994), (float)(0.0003734731), (float)(0.9959107193), (float)(0.0149855051)}};
  for (int g = 0; g < 24; ++g) {
    const float lo = 0.5f * g, hi = 0.5f * (g + 1);
    int base = 0;
    float in = __builtin_huge_valf();
    for (int q = 0; q < 7; ++q) {
      if (lim[q] <= lo) ++base;
      else if (lim[q] < hi) in = lim[q];
    }
    t->lcell[g] = make_float2(in, __int_as_float(base));
    const int up = base + 1 < 8 ? base + 1 : 7;
    t->lone[g][0] = make_float4(k[base][0], k[base][1], k[base][2], k[base][3]);
    t->lone[g][1] = make_float4(k[up][0], k[up][1], k[up][2], k[up][3]);
    t->lone[g][2] = make_float4(in, 0.0f, 0.0f, 0.0f);
  }
  for (int q = 0; q < 8; ++q)
    for (int c = 0; c < 4; ++c) t->lk[c][q] = k[q][c];
  // exp: limits in increasing x; piece p holds x in [elim[p-1], elim[p])
  const float elim[6] = {(float)(-9.91152), (float)(-5.8622823336), (float)(-3.8396630909), (float)(-2.4915033807), (float)(-1.4805375919),
                         (float)(-0.6725053211)};
  const float ec[7][4] = {
      {0.0f, 0.0f, 0.0f, 0.0f},
      {(float)(0.0000803850), (float)(0.0021627428), (float)(0.0194708555), (float)(0.0588080014)},
      {(float)(0.0013889414), (float)(0.0244676474), (float)(0.1471290604), (float)(0.3042757740)},
      {(float)(0.0072335607), (float)(0.0906002677), (float)(0.3983111356), (float)(0.6245959221)},
      {(float)(0.0232410351), (float)(0.2085645908), (float)(0.6906367911), (float)(0.8682322329)},
      {(float)(0.0573782771), (float)(0.3580258429), (float)(0.9121133217), (float)(0.9793091728)},
      {(float)(0.1199175927), (float)(0.4815668234), (float)(0.9975991939), (float)(0.9999505077)}};
  for (int g = 0; g < 20; ++g) {
    const float lo = -10.0f + 0.5f * g, hi = -10.0f + 0.5f * (g + 1);
    int base = 0;
    float in = __builtin_huge_valf();
    for (int q = 0; q < 6; ++q) {
      if (elim[q] <= lo) ++base;
      else if (elim[q] < hi) in = elim[q];
    }
    t->ecell[g] = make_float2(in, __int_as_float(base));
  }
  for (int q = 0; q < 7; ++q)
    for (int c = 0; c < 4; ++c) t->ek[c][q] = ec[q][c];
}

typedef float contra_f2 __attribute__((ext_vector_type(2)));

// two Fast_LogPlusEquals at once: (x.x (+) y.x, x.y (+) y.y); the cubic and the additions on the packed pipe
__device__ __forceinline__ contra_f2 contra_lpe2_t(const contra_tables* t, contra_f2 x, contra_f2 y) {
  contra_f2 hi, lo;
  hi.x = fmaxf(x.x, y.x); hi.y = fmaxf(x.y, y.y);
  lo.x = fminf(x.x, y.x); lo.y = fminf(x.y, y.y);
  const contra_f2 d = hi - lo;
  const contra_f2 d2 = d * 2.0f;
  const unsigned ga = min((unsigned)d2.x, 23u), gb = min((unsigned)d2.y, 23u);
  const float2 ca = t->lcell[ga], cb = t->lcell[gb];
  const unsigned ia = (unsigned)__float_as_int(ca.y) + (d.x >= ca.x ? 1u : 0u), ib = (unsigned)__float_as_int(cb.y) + (d.y >= cb.x ? 1u : 0u);
  const contra_f2 k3 = {t->lk[0][ia], t->lk[0][ib]}, k2 = {t->lk[1][ia], t->lk[1][ib]}, k1 = {t->lk[2][ia], t->lk[2][ib]}, k0 = {t->lk[3][ia], t->lk[3][ib]};
  const contra_f2 r = (((k3 * d + k2) * d + k1) * d + k0) + lo;
  contra_f2 o;
  o.x = (lo.x > (float)(-2e20 / 2) && d.x < (float)(11.8624794162)) ? r.x : hi.x;
  o.y = (lo.y > (float)(-2e20 / 2) && d.y < (float)(11.8624794162)) ? r.y : hi.y;
  return o;
}

// one Fast_LogPlusEquals through the tables
__device__ __forceinline__ float contra_lpe_t(const contra_tables* t, float x, float y) {
  const float hi = fmaxf(x, y), lo = fminf(x, y);
  const float d = hi - lo;
  const unsigned g = min((unsigned)(d * 2.0f), 23u);
  const float2 cl = t->lcell[g];
  const unsigned i = (unsigned)__float_as_int(cl.y) + (d >= cl.x ? 1u : 0u);
  const float r = ((t->lk[0][i] * d + t->lk[1][i]) * d + t->lk[2][i]) * d + t->lk[3][i] + lo;
  return (lo > (float)(-2e20 / 2) && d < (float)(11.8624794162)) ? r : hi;
}

// the same through one lookup (three 16-byte reads of the cell, no dependent second read): for the folding kernel,
// whose single-branch sums are chains of these
__device__ __forceinline__ float contra_lpe_t1(const contra_tables* t, float x, float y) {
  const float hi = fmaxf(x, y), lo = fminf(x, y);
  const float d = hi - lo;
  const unsigned g = min((unsigned)(d * 2.0f), 23u);
  const float4 a = t->lone[g][0], b = t->lone[g][1];
  const float lim = t->lone[g][2].x;
  const bool upper = d >= lim;
  const float k3 = upper ? b.x : a.x, k2 = upper ? b.y : a.y, k1 = upper ? b.z : a.z, k0 = upper ? b.w : a.w;
  const float r = ((k3 * d + k2) * d + k1) * d + k0 + lo;
  return (lo > (float)(-2e20 / 2) && d < (float)(11.8624794162)) ? r : hi;
}

// two Fast_Exp at once (LogSpace.hpp:28-60); arguments >= 0 take the scalar path (expf, see contra_exp)
__device__ __forceinline__ contra_f2 contra_exp2_t(const contra_tables* t, contra_f2 x) {
  const contra_f2 s = (x + 10.0f) * 2.0f;
  const contra_f2 sc = {fmaxf(s.x, 0.0f), fmaxf(s.y, 0.0f)};
  const unsigned ga = min((unsigned)sc.x, 19u), gb = min((unsigned)sc.y, 19u);
  const float2 ca = t->ecell[ga], cb = t->ecell[gb];
  unsigned ia = (unsigned)__float_as_int(ca.y) + (x.x >= ca.x ? 1u : 0u), ib = (unsigned)__float_as_int(cb.y) + (x.y >= cb.x ? 1u : 0u);
  ia = x.x < (float)(-9.91152) ? 0u : ia;  // also covers x < -10, where the cell index was clamped
  ib = x.y < (float)(-9.91152) ? 0u : ib;
  const contra_f2 k3 = {t->ek[0][ia], t->ek[0][ib]}, k2 = {t->ek[1][ia], t->ek[1][ib]}, k1 = {t->ek[2][ia], t->ek[2][ib]}, k0 = {t->ek[3][ia], t->ek[3][ib]};
  contra_f2 r = ((k3 * x + k2) * x + k1) * x + k0;
  if (!(x.x < 0.0f)) r.x = contra_exp(x.x);
  if (!(x.y < 0.0f)) r.y = contra_exp(x.y);
  return r;
}

}  // namespace dafs
