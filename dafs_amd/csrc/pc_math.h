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

}  // namespace dafs
