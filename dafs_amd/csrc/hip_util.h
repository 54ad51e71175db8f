// dafs_amd/csrc/hip_util.h -- small helpers shared by the HIP translation units.
#pragma once
#include <hip/hip_runtime.h>

namespace dafs {

// Records the HIP error text for dafs_hip_last_error(); returns true on failure.
bool hip_check(hipError_t e);

// Dynamic-LDS opt-in of one kernel above the 64 KB default, once per device (the attribute belongs to the device's code
// object: a second context on another GPU of the same process needs its own call).  flags: 16 zero-initialised bools.
inline bool lds_optin_once(const void* fn, int bytes, bool* flags) {
  int dev = 0;
  if (hip_check(hipGetDevice(&dev))) return false;
  if (dev >= 0 && dev < 16 && flags[dev]) return true;
  if (hip_check(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes))) return false;
  if (dev >= 0 && dev < 16) flags[dev] = true;
  return true;
}

#if defined(__HIPCC__)
// Orders LDS accesses of one wavefront: DS operations of a wave execute in issue order, so only
// the compiler has to be kept from moving them.
__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}
#endif

}  // namespace dafs
