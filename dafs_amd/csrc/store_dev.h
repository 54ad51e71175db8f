// dafs_amd/csrc/store_dev.h -- launchers of store_dev.hip (device-resident packing / installing of the sparse stores)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dafs {
int scan_excl_launch(const uint32_t* in, uint32_t mul, uint64_t* out, uint64_t n, hipStream_t st);  // out[n + 1]
int gather_tasks_launch(const uint32_t* task_of_pair, uint64_t p0, uint64_t count, const uint32_t* nnz_by_task, const float* sim_by_task, uint32_t* nnz_out,
                        float* sim_out, hipStream_t st);
int mp_pack_launch(const uint32_t* task_of_pair, uint64_t p0, uint64_t count, const uint64_t* pair_off, const uint32_t* pair_nnz, const uint32_t* col, const float* val,
                   const uint64_t* prefix, uint32_t* col_out, float* val_out, hipStream_t st);
int bp_block_nnz_launch(const uint32_t* rowptr, const uint64_t* blk_rp_off, const uint32_t* blk_len, uint32_t nblk, uint32_t* nnz_by_blk, hipStream_t st);
int bp_by_seq_launch(const uint32_t* seq_of_blk, uint32_t nblk, const uint32_t* nnz_by_blk, const uint64_t* off_by_blk, uint32_t* nnz, uint64_t* bp_off, hipStream_t st);
int bp_pack_launch(uint32_t nseq, const uint64_t* bp_off, const uint32_t* nnz, const uint32_t* col, const float* val, const uint64_t* prefix, uint32_t* col_out,
                   float* val_out, hipStream_t st);
int sim_matrix_launch(const uint32_t* pair_x, const uint32_t* pair_y, const float* sim, uint64_t np, uint32_t n, float* out, hipStream_t st);
}  // namespace dafs
