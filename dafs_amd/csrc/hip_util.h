// dafs_amd/csrc/hip_util.h -- small helpers shared by the HIP translation units.
#pragma once
#include <hip/hip_runtime.h>

namespace dafs {

// Records the HIP error text for dafs_hip_last_error(); returns true on failure.
bool hip_check(hipError_t e);

#if defined(__HIPCC__)
// Orders LDS accesses of one wavefront: DS operations of a wave execute in issue order, so only
// the compiler has to be kept from moving them.
__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}
#endif

}  // namespace dafs
