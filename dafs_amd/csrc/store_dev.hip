// dafs_amd/csrc/store_dev.hip -- the sparse stores in and out of a context WITHOUT leaving the device: what the ranks of a
// multi-GPU run exchange (dist.py: one all-gather of device buffers over RCCL per store).  Export packs a pair-index range
// of a matching-probability store -- whose entry pools are bump-allocated in processing order -- into the canonical layout
// of dafs_hip_mp_fetch (pairs in row-major order: nnz, relative row pointers, entries of mp[x][y] then of mp[y][x]); install
// builds a whole store from such arrays.  The base-pairing store likewise, by sequence.  Index work only: no arithmetic on
// the probabilities.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/dafs_hip.h"
#include "hip_util.h"
#include "store_dev.h"

namespace dafs {

// out[k + 1] = sum of mul * in[0 .. k], out[0] = 0 (one workgroup; n up to a few hundred thousand)
__global__ __launch_bounds__(1024) void k_scan_excl(const uint32_t* __restrict__ in, uint32_t mul, uint64_t* __restrict__ out, uint64_t n) {
  __shared__ uint64_t s_part[1024];
  const uint32_t tid = threadIdx.x, nt = blockDim.x;
  const uint64_t chunk = (n + nt - 1) / nt;
  const uint64_t b = (uint64_t)tid * chunk < n ? (uint64_t)tid * chunk : n, e = b + chunk < n ? b + chunk : n;
  uint64_t sum = 0;
  for (uint64_t k = b; k < e; ++k) sum += (uint64_t)mul * in[k];
  s_part[tid] = sum;
  __syncthreads();
  if (tid == 0) {
    uint64_t run = 0;
    for (uint32_t k = 0; k < nt; ++k) { const uint64_t v = s_part[k]; s_part[k] = run; run += v; }
    out[n] = run;
  }
  __syncthreads();
  uint64_t run = s_part[tid];
  for (uint64_t k = b; k < e; ++k) { out[k] = run; run += (uint64_t)mul * in[k]; }
}

__global__ __launch_bounds__(256) void k_gather_tasks(const uint32_t* __restrict__ task_of_pair, uint64_t p0, uint64_t count, const uint32_t* __restrict__ nnz_by_task,
                                                      const float* __restrict__ sim_by_task, uint32_t* __restrict__ nnz_out, float* __restrict__ sim_out) {
  const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= count) return;
  const uint32_t t = task_of_pair[p0 + k];
  nnz_out[k] = nnz_by_task[t];
  if (sim_out) sim_out[k] = sim_by_task[t];
}

// entries of pair p0 + blockIdx.x: 2 * nnz values from the pool (at pair_off[task]) to their place in pair order
__global__ __launch_bounds__(256) void k_mp_pack(const uint32_t* __restrict__ task_of_pair, uint64_t p0, const uint64_t* __restrict__ pair_off,
                                                 const uint32_t* __restrict__ pair_nnz, const uint32_t* __restrict__ col, const float* __restrict__ val,
                                                 const uint64_t* __restrict__ prefix, uint32_t* __restrict__ col_out, float* __restrict__ val_out) {
  const uint32_t t = task_of_pair[p0 + blockIdx.x];
  const uint64_t src = pair_off[t], dst = prefix[blockIdx.x], n2 = 2ull * pair_nnz[t];
  for (uint64_t e = threadIdx.x; e < n2; e += blockDim.x) { col_out[dst + e] = col[src + e]; val_out[dst + e] = val[src + e]; }
}

// base-pairing store: per sequence x its nnz (last row pointer of its block) -- by block of a gathered layout
__global__ __launch_bounds__(256) void k_bp_block_nnz(const uint32_t* __restrict__ rowptr, const uint64_t* __restrict__ blk_rp_off, const uint32_t* __restrict__ blk_len,
                                                      uint32_t nblk, uint32_t* __restrict__ nnz_by_blk) {
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < nblk) nnz_by_blk[k] = rowptr[blk_rp_off[k] + blk_len[k]];
}
// per sequence: nnz and first entry from its block's
__global__ __launch_bounds__(256) void k_bp_by_seq(const uint32_t* __restrict__ seq_of_blk, uint32_t nblk, const uint32_t* __restrict__ nnz_by_blk,
                                                   const uint64_t* __restrict__ off_by_blk, uint32_t* __restrict__ nnz, uint64_t* __restrict__ bp_off) {
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < nblk) { const uint32_t x = seq_of_blk[k]; nnz[x] = nnz_by_blk[k]; bp_off[x] = off_by_blk[k]; }
}
// entries of sequence blockIdx.x from the pool (at bp_off[x]) to sequence order
__global__ __launch_bounds__(256) void k_bp_pack(const uint64_t* __restrict__ bp_off, const uint32_t* __restrict__ nnz, const uint32_t* __restrict__ col,
                                                 const float* __restrict__ val, const uint64_t* __restrict__ prefix, uint32_t* __restrict__ col_out,
                                                 float* __restrict__ val_out) {
  const uint32_t x = blockIdx.x;
  const uint64_t src = bp_off[x], dst = prefix[x], n = nnz[x];
  for (uint64_t e = threadIdx.x; e < n; e += blockDim.x) { col_out[dst + e] = col[src + e]; val_out[dst + e] = val[src + e]; }
}
// dense N x N similarity matrix with unit diagonal from the per-pair scores in row-major pair order
__global__ __launch_bounds__(256) void k_sim_matrix(const uint32_t* __restrict__ pair_x, const uint32_t* __restrict__ pair_y, const float* __restrict__ sim, uint64_t np,
                                                    uint32_t n, float* __restrict__ out) {
  const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k < np) { const uint32_t x = pair_x[k], y = pair_y[k]; out[(size_t)x * n + y] = sim[k]; out[(size_t)y * n + x] = sim[k]; }
  if (k < n) out[(size_t)k * n + k] = 1.0f;
}

int scan_excl_launch(const uint32_t* in, uint32_t mul, uint64_t* out, uint64_t n, hipStream_t st) {
  hipLaunchKernelGGL(k_scan_excl, dim3(1), dim3(1024), 0, st, in, mul, out, n);
  return hip_check(hipGetLastError()) ? DAFS_HIP_ELAUNCH : DAFS_HIP_OK;
}
int gather_tasks_launch(const uint32_t* task_of_pair, uint64_t p0, uint64_t count, const uint32_t* nnz_by_task, const float* sim_by_task, uint32_t* nnz_out,
                        float* sim_out, hipStream_t st) {
  if (!count) return DAFS_HIP_OK;
  hipLaunchKernelGGL(k_gather_tasks, dim3((uint32_t)((count + 255) / 256)), dim3(256), 0, st, task_of_pair, p0, count, nnz_by_task, sim_by_task, nnz_out, sim_out);
  return hip_check(hipGetLastError()) ? DAFS_HIP_ELAUNCH : DAFS_HIP_OK;
}
int mp_pack_launch(const uint32_t* task_of_pair, uint64_t p0, uint64_t count, const uint64_t* pair_off, const uint32_t* pair_nnz, const uint32_t* col, const float* val,
                   const uint64_t* prefix, uint32_t* col_out, float* val_out, hipStream_t st) {
  if (!count) return DAFS_HIP_OK;
  hipLaunchKernelGGL(k_mp_pack, dim3((uint32_t)count), dim3(256), 0, st, task_of_pair, p0, pair_off, pair_nnz, col, val, prefix, col_out, val_out);
  return hip_check(hipGetLastError()) ? DAFS_HIP_ELAUNCH : DAFS_HIP_OK;
}
int bp_block_nnz_launch(const uint32_t* rowptr, const uint64_t* blk_rp_off, const uint32_t* blk_len, uint32_t nblk, uint32_t* nnz_by_blk, hipStream_t st) {
  if (!nblk) return DAFS_HIP_OK;
  hipLaunchKernelGGL(k_bp_block_nnz, dim3((nblk + 255) / 256), dim3(256), 0, st, rowptr, blk_rp_off, blk_len, nblk, nnz_by_blk);
  return hip_check(hipGetLastError()) ? DAFS_HIP_ELAUNCH : DAFS_HIP_OK;
}
int bp_by_seq_launch(const uint32_t* seq_of_blk, uint32_t nblk, const uint32_t* nnz_by_blk, const uint64_t* off_by_blk, uint32_t* nnz, uint64_t* bp_off, hipStream_t st) {
  if (!nblk) return DAFS_HIP_OK;
  hipLaunchKernelGGL(k_bp_by_seq, dim3((nblk + 255) / 256), dim3(256), 0, st, seq_of_blk, nblk, nnz_by_blk, off_by_blk, nnz, bp_off);
  return hip_check(hipGetLastError()) ? DAFS_HIP_ELAUNCH : DAFS_HIP_OK;
}
int bp_pack_launch(uint32_t nseq, const uint64_t* bp_off, const uint32_t* nnz, const uint32_t* col, const float* val, const uint64_t* prefix, uint32_t* col_out,
                   float* val_out, hipStream_t st) {
  if (!nseq) return DAFS_HIP_OK;
  hipLaunchKernelGGL(k_bp_pack, dim3(nseq), dim3(256), 0, st, bp_off, nnz, col, val, prefix, col_out, val_out);
  return hip_check(hipGetLastError()) ? DAFS_HIP_ELAUNCH : DAFS_HIP_OK;
}
int sim_matrix_launch(const uint32_t* pair_x, const uint32_t* pair_y, const float* sim, uint64_t np, uint32_t n, float* out, hipStream_t st) {
  const uint64_t m = np > n ? np : n;
  hipLaunchKernelGGL(k_sim_matrix, dim3((uint32_t)((m + 255) / 256)), dim3(256), 0, st, pair_x, pair_y, sim, np, n, out);
  return hip_check(hipGetLastError()) ? DAFS_HIP_ELAUNCH : DAFS_HIP_OK;
}

}  // namespace dafs
