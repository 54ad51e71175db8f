#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
template <int Q> __device__ __forceinline__ uint32_t sel_bc(uint32_t n_item, uint32_t t, uint32_t yes, uint32_t no, uint32_t* dd) {
  uint32_t r, d;
  asm volatile("s_nop 1\n\tv_sub_co_u32_dpp %1, vcc, %2, %3 row_newbcast:%6 row_mask:0xf bank_mask:0xf\n\tv_cndmask_b32_e32 %0, %4, %5, vcc"
      : "=v"(r), "=&v"(d) : "v"(n_item), "v"(t), "v"(yes), "v"(no), "n"(Q) : "vcc");
  *dd = d;
  return r;
}
__global__ void k(uint32_t* out, uint32_t* outd, const uint32_t* n) {
  const uint32_t t = (threadIdx.x & 15) + 1;  // t + 1: borrow of n - (t + 1) <=> n <= t <=> inactive
  uint32_t d;
  out[threadIdx.x] = sel_bc<3>(n[threadIdx.x], t, 1u, 0u, &d);
  outd[threadIdx.x] = d;
}
int main() {
  uint32_t hn[64], ho[64], hd[64], *dn, *dout, *dd;
  for (int i = 0; i < 64; ++i) hn[i] = (i & 15) == 3 ? (i >> 4) * 5 + 1 : 100 + i;  // lane 3 of each row: 1, 6, 11, 16
  hipMalloc(&dn, 256); hipMalloc(&dout, 256); hipMalloc(&dd, 256);
  hipMemcpy(dn, hn, 256, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dout, dd, dn);
  hipMemcpy(ho, dout, 256, hipMemcpyDeviceToHost); hipMemcpy(hd, dd, 256, hipMemcpyDeviceToHost);
  for (int r = 0; r < 4; ++r) { printf("row %d (n=%d): ", r, r * 5 + 1); for (int t = 0; t < 16; ++t) printf("%u", ho[r * 16 + t]); printf("  d[0..3]= %d %d %d %d\n", (int)hd[r*16], (int)hd[r*16+1], (int)hd[r*16+2], (int)hd[r*16+3]); }
  return 0;
}
