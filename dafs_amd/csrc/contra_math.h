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

}  // namespace dafs
