// tools/valu_rate.hip -- issue-rate microbenchmark for the instruction kinds the DP kernels are made of (tuning aid).
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/valu_rate tools/valu_rate.hip && tools/valu_rate
//
// For each instruction kind and 1/2/4/8 resident wavefronts per SIMD: shader cycles per wave-instruction per SIMD
// (s_memtime around an unrolled stream of independent instructions, every CU loaded).  Answers what the cost
// models in pair_sweeps.h / DESIGN.md assume: which VALU kinds issue in 2 cycles per wave64 and which in 4 or more,
// and what an LDS lookup costs beside them.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

#define REP 64     // instructions per kind per loop body (8 chains x 8)
#define ITERS 4000

enum { K_ADD, K_MUL, K_MIN, K_CNDMASK, K_PKADD, K_PKMUL, K_ADD64, K_MUL64, K_FMA64, K_CVTFLR, K_LSHL, K_DPP, K_LDS32, K_LDS64, K_LDS128, K_MIX,
       K_FMA, K_MAX, K_AND, K_CND64, K_CMP, K_ADDU, K_LSHLADD, K_MOV, K_MED3, K_CVT64, K_ADDMIN, K_ADDLDS128, K_N };
static const char* kname[K_N] = {"v_add_f32", "v_mul_f32", "v_min_f32", "v_cndmask_b32", "v_pk_add_f32", "v_pk_mul_f32", "v_add_f64", "v_mul_f64", "v_fma_f64",
                                 "v_cvt_flr_i32_f32", "v_lshlrev_b32", "v_mov_b32_dpp", "ds_read_b32", "ds_read_b64", "ds_read_b128", "4 valu + 1 ds_read_b32",
                                 "v_fma_f32", "v_max_f32", "v_and_b32", "v_cndmask_b32_e64 sgpr", "v_cmp_le_f32 -> sgpr", "v_add_u32", "v_lshl_add_u32", "v_mov_b32", "v_med3_f32", "v_cvt_f64_f32",
                                 "v_add_f32 + v_min_f32", "4 v_add_f32 + 1 ds_read_b128"};

typedef float f2 __attribute__((ext_vector_type(2)));

template <int K>
__global__ __launch_bounds__(256) void k_rate(unsigned long long* cycles, float* sink, int iters) {
  __shared__ float lds[4096];
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = (float)i * 0.001f;
  __syncthreads();
  float a[8];
  double d[8];
  f2 p[8];
  int n[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) { a[c] = threadIdx.x * 0.01f + c; d[c] = a[c]; p[c] = f2{a[c], a[c] + 1.0f}; n[c] = threadIdx.x * 7 + c; }
  const float x = 1.0001f + threadIdx.x * 1e-7f;
  const double xd = x;
  const f2 xp = {x, x};
  const unsigned addr0 = (threadIdx.x & 63) * 16;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < REP / 8; ++r) {
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        if (K == K_ADD) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[c]) : "v"(x));
        if (K == K_MUL) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[c]) : "v"(x));
        if (K == K_MIN) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[c]) : "v"(x));
        if (K == K_CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[c]) : "v"(x));
        if (K == K_PKADD) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[c]) : "v"(xp));
        if (K == K_PKMUL) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[c]) : "v"(xp));
        if (K == K_ADD64) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[c]) : "v"(xd));
        if (K == K_MUL64) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[c]) : "v"(xd));
        if (K == K_FMA64) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d[c]) : "v"(xd));
        if (K == K_CVTFLR) asm volatile("v_cvt_flr_i32_f32_e64 %0, -%1" : "=v"(n[c]) : "v"(a[c]));
        if (K == K_LSHL) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(n[c]));
        if (K == K_DPP) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[c]) : "v"(x));
        if (K == K_LDS32) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(a[c]) : "v"(addr0), "n"(0));
        if (K == K_LDS64) asm volatile("ds_read_b64 %0, %1" : "=v"(p[c]) : "v"(addr0));
        if (K == K_LDS128) {
          float4 q;
          asm volatile("ds_read_b128 %0, %1" : "=v"(q) : "v"(addr0));
          a[c] = q.x;
        }
        if (K == K_FMA) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[c]) : "v"(x));
        if (K == K_MAX) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[c]) : "v"(x));
        if (K == K_AND) asm volatile("v_and_b32 %0, %0, %1" : "+v"(n[c]) : "v"(n[(c + 1) & 7]));
        if (K == K_CND64) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[40:41]" : "+v"(a[c]) : "v"(x) : "s40", "s41");
        if (K == K_CMP) asm volatile("v_cmp_le_f32_e64 s[40:41], %0, %1" : : "v"(a[c]), "v"(x) : "s40", "s41");
        if (K == K_ADDU) asm volatile("v_add_u32 %0, %0, %1" : "+v"(n[c]) : "v"(n[(c + 1) & 7]));
        if (K == K_LSHLADD) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(n[c]) : "v"(n[(c + 1) & 7]));
        if (K == K_MOV) asm volatile("v_mov_b32 %0, %1" : "=v"(a[c]) : "v"(x));
        if (K == K_MED3) asm volatile("v_med3_f32 %0, %0, %1, %1" : "+v"(a[c]) : "v"(x));
        if (K == K_CVT64) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[c]) : "v"(a[c]));
        if (K == K_ADDMIN) {
          asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[c]) : "v"(x));
          asm volatile("v_min_f32 %0, %0, %1" : "+v"(p[c].x) : "v"(x));
        }
        if (K == K_ADDLDS128) {
          asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[c]) : "v"(x));
          if ((c & 3) == 0) {
            float4 q;
            asm volatile("ds_read_b128 %0, %1" : "=v"(q) : "v"(addr0));
            n[c] = __float_as_int(q.x);
          }
        }
        if (K == K_MIX) {
          asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[c]) : "v"(x));
          if ((c & 3) == 0) asm volatile("ds_read_b32 %0, %1" : "=v"(n[c]) : "v"(addr0));
        }
      }
    }
    if ((K >= K_LDS32 && K <= K_MIX) || K == K_ADDLDS128) asm volatile("s_waitcnt lgkmcnt(0)");
  }
  asm volatile("s_waitcnt lgkmcnt(0)");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
#pragma unroll
  for (int c = 0; c < 8; ++c) s += a[c] + (float)d[c] + p[c].x + p[c].y + (float)n[c];
  if (s == 12345.678f) sink[0] = s;
  if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int K>
static void run(int waves_per_simd, unsigned long long* d_cyc, float* d_sink, int cus) {
  // one 256-thread workgroup = one wavefront per SIMD of a CU; waves_per_simd workgroups per CU
  const int blocks = cus * waves_per_simd;
  hipLaunchKernelGGL(k_rate<K>, dim3(blocks), dim3(256), 0, 0, d_cyc, d_sink, 10);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(k_rate<K>, dim3(blocks), dim3(256), 0, 0, d_cyc, d_sink, ITERS);
  hipEventRecord(e1, 0);
  hipDeviceSynchronize();
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(blocks * 4);
  hipMemcpy(h.data(), d_cyc, h.size() * sizeof h[0], hipMemcpyDeviceToHost);
  std::sort(h.begin(), h.end());
  const double med = (double)h[h.size() / 2];
  const double per_wave_instr = (K == K_MIX || K == K_ADDLDS128) ? (double)REP * 1.25 : (K == K_ADDMIN ? (double)REP * 2 : (double)REP);
  // cycles the SIMD spends per wave-instruction = wave time / (instructions per wave * waves sharing the SIMD)
  // wall clock: ns the SIMD spends per wave-instruction (launch overhead included: ITERS is large)
  printf(" | %dw %5.2f (%5.2f) %5.2fns", waves_per_simd, med / (per_wave_instr * ITERS * waves_per_simd), med / (per_wave_instr * ITERS),
         ms * 1e6 / (per_wave_instr * ITERS * waves_per_simd));
}

template <int K>
static void all(unsigned long long* d_cyc, float* d_sink, int cus) {
  printf("%-24s", kname[K]);
  for (int w : {1, 2, 3, 4, 5, 6, 8}) run<K>(w, d_cyc, d_sink, cus);
  printf("\n");
}

int main() {
  hipDeviceProp_t pr;
  if (hipGetDeviceProperties(&pr, 0) != hipSuccess) { fprintf(stderr, "no HIP device\n"); return 1; }
  const int cus = pr.multiProcessorCount;
  printf("%s, %d CUs; per kind and resident wavefronts per SIMD: s_memtime cycles per wave-instruction per SIMD (cycles between two instructions of one wave)\n", pr.name, cus);
  unsigned long long* d_cyc;
  float* d_sink;
  hipMalloc(&d_cyc, sizeof(unsigned long long) * cus * 8 * 4);
  hipMalloc(&d_sink, 64);
  all<K_ADD>(d_cyc, d_sink, cus); all<K_MUL>(d_cyc, d_sink, cus); all<K_MIN>(d_cyc, d_sink, cus); all<K_CNDMASK>(d_cyc, d_sink, cus);
  all<K_PKADD>(d_cyc, d_sink, cus); all<K_PKMUL>(d_cyc, d_sink, cus); all<K_ADD64>(d_cyc, d_sink, cus); all<K_MUL64>(d_cyc, d_sink, cus);
  all<K_FMA64>(d_cyc, d_sink, cus); all<K_CVTFLR>(d_cyc, d_sink, cus); all<K_LSHL>(d_cyc, d_sink, cus); all<K_DPP>(d_cyc, d_sink, cus);
  all<K_LDS32>(d_cyc, d_sink, cus); all<K_LDS64>(d_cyc, d_sink, cus); all<K_LDS128>(d_cyc, d_sink, cus); all<K_MIX>(d_cyc, d_sink, cus);
  all<K_FMA>(d_cyc, d_sink, cus); all<K_MAX>(d_cyc, d_sink, cus); all<K_AND>(d_cyc, d_sink, cus); all<K_CND64>(d_cyc, d_sink, cus); all<K_CMP>(d_cyc, d_sink, cus);
  all<K_ADDU>(d_cyc, d_sink, cus); all<K_LSHLADD>(d_cyc, d_sink, cus); all<K_MOV>(d_cyc, d_sink, cus); all<K_MED3>(d_cyc, d_sink, cus); all<K_CVT64>(d_cyc, d_sink, cus);
  all<K_ADDMIN>(d_cyc, d_sink, cus); all<K_ADDLDS128>(d_cyc, d_sink, cus);
  return 0;
}
