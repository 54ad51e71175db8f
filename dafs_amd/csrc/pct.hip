// dafs_amd/csrc/pct.hip -- probabilistic consistency transforms on the sparse posterior stores.
//
//   k_pct_match : DAFS::relax_matching_probability      reference src/dafs.cpp:258-324
//   k_pct_bp    : DAFS::relax_basepairing_probability   reference src/dafs.cpp:326-375
//
// Both are sparse triple loops that accumulate float products into a dense per-output matrix in a
// fixed order (z, then k ascending).  Bit-exact parity needs every cell to receive its addends in
// that order, so the work is split by OUTPUT ROW: one workgroup owns one output matrix, keeps a
// tile of it in LDS, and each thread owns whole rows of the tile -- it walks the reference's
// (z, k) order itself, so no atomics and no reduction tree are involved.  The tile is as many rows
// as fit in LDS; when an output needs several tiles the transform is computed twice (count,
// reserve pool space, then recompute and emit), single-tile outputs emit straight from LDS.
// Output: rows with v > CUTOFF (0.01) as CSR, plus the transposed CSR for the matching matrices,
// bump-allocated from a pool exactly like k_pairhmm3's.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/dafs_hip.h"
#include "hip_util.h"
#include "sparse_view.h"
#include "pct.h"

namespace dafs {

#define PCT_CUTOFF 0.01f  // reference CUTOFF is the double 0.01; for a float v, v > 0.01 <=> v > 0.01f

__global__ __launch_bounds__(256) void k_pct_match(pct_match_args a) {
  extern __shared__ float s_mem[];
  const uint32_t N = a.in.nseq;
  const uint32_t tid = threadIdx.x, nt = blockDim.x;
  float* wz = s_mem;                                      // N
  uint32_t* rowptr = (uint32_t*)(wz + N);                 // max_len + 2
  uint32_t* colptr = rowptr + a.max_len + 2;              // max_len + 2
  uint32_t* colcur = colptr + a.max_len + 2;              // max_len + 2
  float* acc = (float*)(colcur + a.max_len + 2);          // tile_cells
  __shared__ float s_sum_w;
  __shared__ unsigned long long s_off;
  __shared__ int s_ok;

  for (uint32_t p = blockIdx.x; p < a.npairs; p += gridDim.x) {
    const uint32_t x = a.pair_x[p], y = a.pair_y[p];
    const uint32_t L1 = a.in.len[x], L2 = a.in.len[y];
    // dafs.cpp:280-288
    for (uint32_t z = tid; z < N; z += nt) {
      float w = a.sim[(size_t)z * N + x] * a.sim[(size_t)z * N + y];
      if (a.w_pct < 0.0) w *= 1.0 / N;
      else if (z == x || z == y) w *= (1.0 - a.w_pct) / 2;
      else w *= a.w_pct / (N - 2);
      wz[z] = w;
    }
    for (uint32_t j = tid; j <= L2; j += nt) { colptr[j] = 0; colcur[j] = 0; }
    __syncthreads();
    if (tid == 0) {
      float s = 0.0f;
      for (uint32_t z = 0; z < N; ++z) s += wz[z];
      s_sum_w = s;
    }
    __syncthreads();
    const float sum_w = s_sum_w;
    const uint32_t TI = min(L1, a.tile_cells / L2);
    const uint32_t ntiles = (L1 + TI - 1) / TI;
    const int npass = ntiles == 1 ? 1 : 2;
    unsigned long long off = 0;
    uint32_t nnz = 0;
    bool ok = true;

    for (int pass = 0; pass < npass; ++pass) {
      const bool last = pass == npass - 1;
      for (uint32_t tile = 0; tile < ntiles; ++tile) {
        const uint32_t i0 = tile * TI;
        const uint32_t rows = min(TI, L1 - i0);
        for (uint32_t c = tid; c < rows * L2; c += nt) acc[c] = 0.0f;
        __syncthreads();
        for (uint32_t r = tid; r < rows; r += nt) {
          const uint32_t i = i0 + r;
          float* arow = acc + (size_t)r * L2;
          for (uint32_t z = 0; z < N; ++z) {
            const float w = wz[z];
            if (z == x) {  // mp[x][x][k] = {(k,1)}: only k == i reaches row i
              const row_ref b = mp_row(a.in, x, y, i);
              for (uint32_t e = 0; e < b.n; ++e) arow[b.col[e]] += 1.0f * b.val[e] * w;
            } else if (z == y) {  // mp[y][y][k] = {(k,1)}
              const row_ref ar = mp_row(a.in, x, y, i);
              for (uint32_t e = 0; e < ar.n; ++e) arow[ar.col[e]] += ar.val[e] * 1.0f * w;
            } else {
              const row_ref ar = mp_row(a.in, x, z, i);  // (k, p_ik), k ascending
              for (uint32_t ea = 0; ea < ar.n; ++ea) {
                const float p_ik = ar.val[ea];
                const row_ref b = mp_row(a.in, z, y, ar.col[ea]);  // (j, p_jk)
                for (uint32_t eb = 0; eb < b.n; ++eb) arow[b.col[eb]] += p_ik * b.val[eb] * w;
              }
            }
          }
        }
        __syncthreads();
        if (pass == 0) {  // count rows and columns of this tile
          for (uint32_t r = tid; r < rows; r += nt) {
            uint32_t c = 0;
            for (uint32_t j = 0; j < L2; ++j) c += (acc[(size_t)r * L2 + j] / sum_w > PCT_CUTOFF) ? 1 : 0;
            rowptr[i0 + r + 1] = c;
          }
          for (uint32_t j = tid; j < L2; j += nt) {
            uint32_t c = 0;
            for (uint32_t r = 0; r < rows; ++r) c += (acc[(size_t)r * L2 + j] / sum_w > PCT_CUTOFF) ? 1 : 0;
            colptr[j + 1] += c;
          }
          __syncthreads();
        }
        if (pass == 0 && tile == ntiles - 1) {  // all counts known: prefix sums + pool reservation
          if (tid == 0) {
            rowptr[0] = 0;
            for (uint32_t i = 0; i < L1; ++i) rowptr[i + 1] += rowptr[i];
            colptr[0] = 0;
            for (uint32_t j = 0; j < L2; ++j) colptr[j + 1] += colptr[j];
            const uint32_t n = rowptr[L1];
            const unsigned long long o = atomicAdd(a.pool_top, 2ull * n);
            s_off = o;
            s_ok = (o + 2ull * n <= a.pool_cap) ? 1 : 0;
            if (!s_ok) atomicExch(a.status, DAFS_HIP_EOVERFLOW);
            a.pair_off[p] = o;
            a.pair_nnz[p] = n;
          }
          __syncthreads();
          off = s_off;
          ok = s_ok != 0;
          nnz = rowptr[L1];
          const uint64_t rp = a.rp_off[p];
          for (uint32_t i = tid; i <= L1; i += nt) a.rowptr_pool[rp + i] = rowptr[i];
          for (uint32_t j = tid; j <= L2; j += nt) a.rowptr_pool[rp + L1 + 1 + j] = colptr[j];
        }
        if (last && ok) {  // emit this tile: rows by row-threads, columns by column-threads
          for (uint32_t r = tid; r < rows; r += nt) {
            unsigned long long pos = off + rowptr[i0 + r];
            for (uint32_t j = 0; j < L2; ++j) {
              const float v = acc[(size_t)r * L2 + j] / sum_w;
              if (v > PCT_CUTOFF) { a.col[pos] = j; a.val[pos] = v; ++pos; }
            }
          }
          for (uint32_t j = tid; j < L2; j += nt) {
            unsigned long long pos = off + nnz + colptr[j] + colcur[j];
            uint32_t c = 0;
            for (uint32_t r = 0; r < rows; ++r) {
              const float v = acc[(size_t)r * L2 + j] / sum_w;
              if (v > PCT_CUTOFF) { a.col[pos] = i0 + r; a.val[pos] = v; ++pos; ++c; }
            }
            colcur[j] += c;
          }
        }
        __syncthreads();
      }
    }
  }
}

__global__ __launch_bounds__(256) void k_pct_bp(pct_bp_args a) {
  extern __shared__ float s_mem[];
  const uint32_t N = a.mp.nseq;
  const uint32_t tid = threadIdx.x, nt = blockDim.x;
  float* wy = s_mem;                              // N
  uint32_t* rowptr = (uint32_t*)(wy + N);         // max_len + 2
  float* acc = (float*)(rowptr + a.max_len + 2);  // tile_cells
  __shared__ float s_sum_w;
  __shared__ unsigned long long s_off;
  __shared__ int s_ok;

  for (uint32_t x = blockIdx.x; x < N; x += gridDim.x) {
    const uint32_t L1 = a.mp.len[x];
    // dafs.cpp:341-348
    for (uint32_t y = tid; y < N; y += nt) {
      float w = a.sim[(size_t)y * N + x];
      if (a.w_pct < 0.0) w *= 1.0 / N;
      else if (y == x) w *= 1.0 - a.w_pct;
      else w *= a.w_pct / (N - 1);
      wy[y] = w;
    }
    __syncthreads();
    if (tid == 0) {
      float s = 0.0f;
      for (uint32_t y = 0; y < N; ++y) s += wy[y];
      s_sum_w = s;
    }
    __syncthreads();
    const float sum_w = s_sum_w;
    const uint32_t TI = min(L1, a.tile_cells / L1);
    const uint32_t ntiles = (L1 + TI - 1) / TI;
    const int npass = ntiles == 1 ? 1 : 2;
    unsigned long long off = 0;
    bool ok = true;

    for (int pass = 0; pass < npass; ++pass) {
      const bool last = pass == npass - 1;
      for (uint32_t tile = 0; tile < ntiles; ++tile) {
        const uint32_t i0 = tile * TI;
        const uint32_t rows = min(TI, L1 - i0);
        for (uint32_t c = tid; c < rows * L1; c += nt) acc[c] = 0.0f;
        __syncthreads();
        for (uint32_t r = tid; r < rows; r += nt) {
          const uint32_t i = i0 + r;
          float* arow = acc + (size_t)r * L1;
          for (uint32_t y = 0; y < N; ++y) {
            const float w = wy[y];
            if (y == x) {  // mp[x][x] is the identity: k = i, p_ik = 1; j = l, p_jl = 1
              const row_ref b = bp_row(a.bp, x, i);
              for (uint32_t e = 0; e < b.n; ++e) {
                const uint32_t j = b.col[e];
                if (i < j) arow[j] += b.val[e] * 1.0f * 1.0f * w;
              }
            } else {
              const row_ref ar = mp_row(a.mp, x, y, i);  // (k, p_ik) = entries (i, p_ik) of mp[y][x][k], k ascending
              for (uint32_t ea = 0; ea < ar.n; ++ea) {
                const uint32_t k = ar.col[ea];
                const float p_ik = ar.val[ea];
                const row_ref b = bp_row(a.bp, y, k);  // (l, p_kl)
                for (uint32_t eb = 0; eb < b.n; ++eb) {
                  const float p_kl = b.val[eb];
                  const row_ref c = mp_row(a.mp, y, x, b.col[eb]);  // mp[y][x][l]: (j, p_jl)
                  for (uint32_t ec = 0; ec < c.n; ++ec) {
                    const uint32_t j = c.col[ec];
                    if (i < j) arow[j] += p_kl * p_ik * c.val[ec] * w;
                  }
                }
              }
            }
          }
        }
        __syncthreads();
        if (pass == 0) {
          for (uint32_t r = tid; r < rows; r += nt) {
            uint32_t c = 0;
            for (uint32_t j = i0 + r + 1; j < L1; ++j) c += (acc[(size_t)r * L1 + j] / sum_w > PCT_CUTOFF) ? 1 : 0;
            rowptr[i0 + r + 1] = c;
          }
          __syncthreads();
        }
        if (pass == 0 && tile == ntiles - 1) {
          if (tid == 0) {
            rowptr[0] = 0;
            for (uint32_t i = 0; i < L1; ++i) rowptr[i + 1] += rowptr[i];
            const uint32_t n = rowptr[L1];
            const unsigned long long o = atomicAdd(a.pool_top, (unsigned long long)n);
            s_off = o;
            s_ok = (o + n <= a.pool_cap) ? 1 : 0;
            if (!s_ok) atomicExch(a.status, DAFS_HIP_EOVERFLOW);
            a.out_off[x] = o;
            a.out_nnz[x] = n;
          }
          __syncthreads();
          off = s_off;
          ok = s_ok != 0;
          const uint64_t rp = a.bp.rp_off[x];  // same row-pointer layout as the input store
          for (uint32_t i = tid; i <= L1; i += nt) a.out_rowptr[rp + i] = rowptr[i];
        }
        if (last && ok) {
          for (uint32_t r = tid; r < rows; r += nt) {
            unsigned long long pos = off + rowptr[i0 + r];
            for (uint32_t j = i0 + r + 1; j < L1; ++j) {
              const float v = acc[(size_t)r * L1 + j] / sum_w;
              if (v > PCT_CUTOFF) { a.out_col[pos] = j; a.out_val[pos] = v; ++pos; }
            }
          }
        }
        __syncthreads();
      }
    }
  }
}

static const size_t kPctLdsBytes = 150 * 1024;  // leave room for the static LDS and alignment

int pct_match_launch(pct_match_args a, uint32_t max_len, hipStream_t st) {
  const size_t fixed = ((size_t)a.in.nseq + 3 * ((size_t)max_len + 2)) * 4;
  if (fixed + (size_t)max_len * 4 > kPctLdsBytes) return DAFS_HIP_ETOOLONG;
  a.max_len = max_len;
  a.tile_cells = (uint32_t)((kPctLdsBytes - fixed) / 4);
  static bool attr = false;
  if (!attr) {
    if (hip_check(hipFuncSetAttribute((const void*)k_pct_match, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kPctLdsBytes))) return DAFS_HIP_ELAUNCH;
    attr = true;
  }
  const uint32_t grid = a.npairs < 4096 ? a.npairs : 4096;
  hipLaunchKernelGGL(k_pct_match, dim3(grid), dim3(256), kPctLdsBytes, st, a);
  return hip_check(hipGetLastError()) ? DAFS_HIP_ELAUNCH : DAFS_HIP_OK;
}

int pct_bp_launch(pct_bp_args a, uint32_t max_len, hipStream_t st) {
  const size_t fixed = ((size_t)a.mp.nseq + ((size_t)max_len + 2)) * 4;
  if (fixed + (size_t)max_len * 4 > kPctLdsBytes) return DAFS_HIP_ETOOLONG;
  a.max_len = max_len;
  a.tile_cells = (uint32_t)((kPctLdsBytes - fixed) / 4);
  static bool attr = false;
  if (!attr) {
    if (hip_check(hipFuncSetAttribute((const void*)k_pct_bp, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kPctLdsBytes))) return DAFS_HIP_ELAUNCH;
    attr = true;
  }
  hipLaunchKernelGGL(k_pct_bp, dim3(a.mp.nseq), dim3(256), kPctLdsBytes, st, a);
  return hip_check(hipGetLastError()) ? DAFS_HIP_ELAUNCH : DAFS_HIP_OK;
}

}  // namespace dafs
