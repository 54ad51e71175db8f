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
#include "stage.h"
#include "sparse_view.h"
#include "pct.h"

namespace dafs {

#define PCT_CUTOFF 0.01f  // reference CUTOFF is the double 0.01; for a float v, v > 0.01 <=> v > 0.01f

// ---------------------------------------------------------------------------------------------
// relax_matching_probability in two kernels.
//
// k_pct_rows: one wavefront per output row i of a pair (x,y).  The addends of the row are
//   mp[x][z][i][k] * mp[z][y][k][j] * w_z, added to cell j in the order z, k, j (dafs.cpp:290-318).
//   For a chunk of 32 sequences z the wave first fetches all rows mp[x][z][i] (one lane per z), then
//   the row pointers of every mp[z][y][k] they reference (one lane per (z,k) "item"), then the entries
//   of those rows (one lane per addend), so the pointer chases of a whole chunk overlap; the addends
//   are applied item by item -- the columns of one item are distinct, so its lanes update the row
//   accumulator (LDS) together, and consecutive items follow each other in program order.  The
//   finished row goes to a dense tile in HBM.
// k_pct_emit: one workgroup per pair turns the tile into the thresholded CSR and transposed CSR
//   (rows by wavefronts with ballot compaction, columns by threads), bump-allocated like k_pairhmm3's.
// ---------------------------------------------------------------------------------------------
#define PCT_G 16            // lanes per output row
#define PCT_ROWS_PER_WG 16  // 256 threads / PCT_G

template <int Q>
__device__ __forceinline__ uint32_t row_bcast(uint32_t v) {  // lane Q of every 16-lane row, to the whole row
  return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x150 + Q, 0xF, 0xF, false);  // every lane has a source: no `old` operand to set up
}
template <int Q>
__device__ __forceinline__ float row_bcastf(float v) { return __uint_as_float(row_bcast<Q>(__float_as_uint(v))); }

typedef uint32_t pct_u2 __attribute__((ext_vector_type(2)));  // (a builtin vector: HIP's uint2 class cannot be read through an address-space pointer)
typedef __attribute__((address_space(1))) const pct_u2 pct_gent;  // entries are in global memory: global_load, not flat_load
struct pct_item {  // one (z, k) of a row, held by one lane of the row's group
  uint32_t ptr_lo, ptr_hi;  // address of the first entry of mp[z][y][k] in the interleaved pool (a byte address: one 64-bit add per fetch)
  uint32_t n;               // its entries; an identity row (z == y) is the one entry {k, 1} of ident2
  float pik, w;
};

// Items Q..15 of a round: phase A fetches the first 16 entries of every item's b-row (all loads in flight
// together), phase B applies the items in order.  Entries of one item have distinct columns, so the lanes
// of a group update the row accumulator together; the groups of a wavefront are different rows.
// which of the 16 item slots of a round hold, in some group of the wavefront, a b-row longer than a group (wave-uniform)
__device__ __forceinline__ uint32_t pct_long_slots(uint32_t n) {
  const unsigned long long m = __ballot(n > 16u);
  return (uint32_t)((m | (m >> 16) | (m >> 32) | (m >> 48)) & 0xFFFFull);
}

template <int Q>
struct pct_round {
  static __device__ __forceinline__ void fetch(const uint2* __restrict__ ent, const pct_item& it, int t, uint32_t (&jc)[16], float (&pv)[16]) {
    const uint32_t n = row_bcast<Q>(it.n), lo = row_bcast<Q>(it.ptr_lo), hi = row_bcast<Q>(it.ptr_hi);
    // column and value in one 8-byte load; the pair leaves the branch as loaded (taking it apart inside would make every
    // load wait for itself) and is taken apart after the last one has been issued
    pct_u2 cv = {lo, hi};  // lanes without an entry keep what is in the registers anyway (never used: apply tests t < n)
    if ((uint32_t)t < n) cv = ((const pct_gent*)(((uint64_t)hi << 32) | lo))[t];
    pct_round<Q + 1>::fetch(ent, it, t, jc, pv);
    jc[Q] = cv.x; pv[Q] = __uint_as_float(cv.y);
  }
  // jlo: only columns >= jlo take part (the base-pair transform keeps i < j)
  static __device__ __forceinline__ void apply(const uint2* __restrict__ ent, const pct_item& it, int t, uint32_t jlo, const uint32_t (&jc)[16],
                                               const float (&pv)[16], float* acc, uint32_t longq) {
    const uint32_t n = row_bcast<Q>(it.n);
    const float pik = row_bcastf<Q>(it.pik), w = row_bcastf<Q>(it.w);
    if ((uint32_t)t < n && jc[Q] >= jlo) acc[jc[Q]] += pik * pv[Q] * w;  // dafs.cpp:300 / :308 / :316, :359 / :368
    if (longq & (1u << Q)) {  // (scalar) a b-row longer than the group: the rest of it, before the next item
      const uint32_t lo = row_bcast<Q>(it.ptr_lo), hi = row_bcast<Q>(it.ptr_hi);
      const pct_gent* base = (const pct_gent*)(((uint64_t)hi << 32) | lo);
      uint32_t nmax = n;
#pragma unroll
      for (int o = 16; o < 64; o <<= 1) nmax = max(nmax, (uint32_t)__shfl_xor((int)nmax, o));
      for (uint32_t e0 = 16; e0 < nmax; e0 += 16) {
        wave_lds_fence();
        if (e0 + (uint32_t)t < n) {
          const pct_u2 cv = base[e0 + (uint32_t)t];
          if (cv.x >= jlo) acc[cv.x] += pik * __uint_as_float(cv.y) * w;
        }
      }
    }
    wave_lds_fence();
    pct_round<Q + 1>::apply(ent, it, t, jlo, jc, pv, acc, longq);
  }
};
template <>
struct pct_round<16> {
  static __device__ __forceinline__ void fetch(const uint2*, const pct_item&, int, uint32_t (&)[16], float (&)[16]) {}
  static __device__ __forceinline__ void apply(const uint2*, const pct_item&, int, uint32_t, const uint32_t (&)[16], const float (&)[16], float*, uint32_t) {}
};

// ---- round 3: the lean step ----------------------------------------------------------------------------------------
// k_pct_rows is bound by vector issue (profiles/r02_j: 4.7 G VALU wave-instructions for 6.1 G addends) and the steps of
// a round were 14 VALU instructions each: three broadcasts for the fetch, three for the apply, two compares with their
// exec juggling, 64-bit address arithmetic.  Here a step is 8:
//   fetch  address of entry t = item's byte address + 8 t as v_add_co / v_addc_co with the DPP broadcast folded into the
//          operand (no v_mov_dpp); the load is NOT predicated -- lanes beyond the item's length read the entries that follow
//          in the pool (the pools are padded by 16 entries), which costs no additional cache line
//   apply  column = item length > t ? loaded column : the lane's own trash cell behind the row (one v_cmp_dpp + v_cndmask:
//          no exec change, hence no exec-write -> DPP hazard stalls), two v_mul_f32_dpp (p_ik and w_z arrive through the DPP
//          operand), address, LDS read, add, LDS write.
// The DPP operands are inline assembly (the compiler does not fold v_mov_dpp row_newbcast into its users); "s_nop 1" in
// front keeps the VALU-write -> DPP-read distance whatever the scheduler puts before the block.
template <int Q>
__device__ __forceinline__ float pct_mul_bc(float item, float x) {  // (item value of lane Q of the row) * x
  float r;
  asm("s_nop 1\n\tv_mul_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(item), "v"(x), "n"(Q));
  return r;
}
template <int Q>
__device__ __forceinline__ uint32_t pct_sel_bc(uint32_t n_item, uint32_t t1, uint32_t yes, uint32_t no) {  // n of lane Q > t1 - 1 ? yes : no
  // VOPC has no DPP form on gfx9, and the DPP operand of a subtraction is always the minuend (measured: v_subrev_dpp swaps
  // before the DPP fetch, tools/scratch/dpp_test.hip): the borrow of n - (t + 1) says n <= t, i.e. the lane has no entry
  uint32_t r, d;
  asm("s_nop 1\n\tv_sub_co_u32_dpp %1, vcc, %2, %3 row_newbcast:%6 row_mask:0xf bank_mask:0xf\n\tv_cndmask_b32_e32 %0, %4, %5, vcc"
      : "=v"(r), "=&v"(d) : "v"(n_item), "v"(t1), "v"(yes), "v"(no), "n"(Q) : "vcc");
  return r;
}
template <int Q>
__device__ __forceinline__ uint64_t pct_addr_bc(uint32_t lo_item, uint32_t hi_item, uint32_t t8, uint32_t zero) {  // address of lane Q + t8
  uint32_t lo, hi;
  asm("s_nop 1\n\tv_add_co_u32_dpp %0, vcc, %2, %4 row_newbcast:%6 row_mask:0xf bank_mask:0xf\n\t"
      "v_addc_co_u32_dpp %1, vcc, %3, %5, vcc row_newbcast:%6 row_mask:0xf bank_mask:0xf"
      : "=&v"(lo), "=v"(hi) : "v"(lo_item), "v"(hi_item), "v"(t8), "v"(zero), "n"(Q) : "vcc");
  return ((uint64_t)hi << 32) | lo;
}

#ifndef PCT_TRASH_CELLS
#define PCT_TRASH_CELLS 0
#endif
#ifndef PCT_FETCH_ALL
#define PCT_FETCH_ALL 1  // 1: unpredicated fetch (two VALU instructions fewer per step; 0, predicated, measures the same: the kernel is issue-bound, not line-bound)
#endif
template <int Q>
struct pct_step {
  static __device__ __forceinline__ void fetch(const pct_item& it, uint32_t t8, uint32_t zero, uint32_t trash, pct_u2 (&cv)[16]) {
#if PCT_FETCH_ALL
    cv[Q] = *(const pct_gent*)pct_addr_bc<Q>(it.ptr_lo, it.ptr_hi, t8, zero);
#else
    // only the lanes that have an entry load: a 16-lane span behind a row's start crosses a 128-byte line far more often than
    // the row's own ~6 entries do, and the kernel is bound by the line requests its gathers make
    const uint64_t adr = pct_addr_bc<Q>(it.ptr_lo, it.ptr_hi, t8, zero);
    pct_u2 e = {trash, 0u};
    if (pct_sel_bc<Q>(it.n, (t8 >> 3) + 1u, 1u, 0u)) e = *(const pct_gent*)adr;
    cv[Q] = e;
#endif
    pct_step<Q + 1>::fetch(it, t8, zero, trash, cv);
  }
  static __device__ __forceinline__ void apply(const pct_item& it, uint32_t t, uint32_t trash, const pct_u2 (&cv)[16], float* acc, uint32_t longq) {
    const float v = pct_mul_bc<Q>(it.w, pct_mul_bc<Q>(it.pik, __uint_as_float(cv[Q].y)));  // (p_ik * p_kj) * w_z, dafs.cpp:300 / :308 / :316
#if PCT_TRASH_CELLS
    // every lane updates a cell, the ones without an entry their own trash cell: no exec change, but all 64 lanes go through
    // the LDS, and the LDS is what binds (profiles/r03_a_pct_pmc.json: its pipe is busy 70 % of the kernel, 40 % of that conflicts)
#if PCT_FETCH_ALL
    const uint32_t col = pct_sel_bc<Q>(it.n, t + 1u, cv[Q].x, trash);
#else
    const uint32_t col = cv[Q].x;  // lanes without an entry carry {their trash cell, 0} from the fetch
#endif
    acc[col] += v;
#else
    if (pct_sel_bc<Q>(it.n, t + 1u, 1u, 0u)) acc[cv[Q].x] += v;  // only the lanes with an entry touch the LDS
#endif
    if (longq & (1u << Q)) {  // (scalar, rare) a b-row longer than the group: the rest of it, before the next item
      const uint32_t n = row_bcast<Q>(it.n), lo = row_bcast<Q>(it.ptr_lo), hi = row_bcast<Q>(it.ptr_hi);
      const float pik = row_bcastf<Q>(it.pik), w = row_bcastf<Q>(it.w);
      const pct_gent* base = (const pct_gent*)(((uint64_t)hi << 32) | lo);
      uint32_t nmax = n;
#pragma unroll
      for (int o = 16; o < 64; o <<= 1) nmax = max(nmax, (uint32_t)__shfl_xor((int)nmax, o));
      for (uint32_t e0 = 16; e0 < nmax; e0 += 16) {
        wave_lds_fence();
        if (e0 + t < n) {
          const pct_u2 e = base[e0 + t];
          acc[e.x] += pik * __uint_as_float(e.y) * w;
        }
      }
    }
    wave_lds_fence();
    pct_step<Q + 1>::apply(it, t, trash, cv, acc, longq);
  }
};
template <>
struct pct_step<16> {
  static __device__ __forceinline__ void fetch(const pct_item&, uint32_t, uint32_t, uint32_t, pct_u2 (&)[16]) {}
  static __device__ __forceinline__ void apply(const pct_item&, uint32_t, uint32_t, const pct_u2 (&)[16], float*, uint32_t) {}
};

// LDS of one group (= one output row), in words: 16 trash cells behind the accumulator row, then the chunk's bookkeeping
// (item address bases 16 x 2, exclusive item offsets 20; 16 spare).  row_cap = 28 (mod 32) makes the stride 16 (mod 32):
// the two rows a 32-lane half updates sit sixteen banks apart (see pct_match_launch).
#define PCT_TRASH 16
#define PCT_BOOK_WORDS (16 * 2 + 20 + 16)
__host__ __device__ static inline size_t pct_group_words(uint32_t row_cap) { return (size_t)row_cap + PCT_TRASH + PCT_BOOK_WORDS; }

__global__ __launch_bounds__(256) void k_pct_rows(pct_match_args a, uint32_t pair0, uint32_t row_cap) {
  extern __shared__ unsigned char s_raw[];
  const uint32_t N = a.in.nseq;
  const uint32_t tid = threadIdx.x, nt = blockDim.x;
  const uint32_t t = tid & 15u;
  const int gw = (int)(tid >> 4);
  // workgroup tables, per z: ADDRESSES of the row pointers and of the entries of mp[x][z] (a-side) and of mp[z][y] (b-side),
  // and w_z.  The identity matrices mp[x][x] and mp[y][y] (align.cpp:42-44) are ordinary CSRs here (ident_rp / ident2), so
  // that no step of the row loop knows about them.
  uint64_t* a_rp = (uint64_t*)s_raw;
  uint64_t* a_ent = a_rp + N;
  uint64_t* b_rp = a_ent + N;
  uint64_t* b_ent = b_rp + N;
  float* wz = (float*)(b_ent + N);
  // per group (= per output row): the accumulator row, the lanes' trash cells, the a-row bookkeeping of the current chunk
  float* acc = wz + ((N + 1) & ~1u) + (size_t)gw * pct_group_words(row_cap);
  uint64_t* zabase = (uint64_t*)(acc + row_cap + PCT_TRASH);  // per z-lane: address of item tl of the chunk = zabase + 8 tl
  uint32_t* zoff = (uint32_t*)(zabase + 16);                   // exclusive item offsets of the z-lanes (zoff[16] = items of the chunk)

  // which (pair, block of 16 rows) this workgroup takes: from the launch's task table (pct_match_launch: ordered so that the
  // workgroups an XCD runs at the same time share y and the row block, i.e. the b-rows they gather), or from the 2-D grid
  uint32_t pl = blockIdx.x, rb = blockIdx.y;
  if (a.wg_task) {
    const uint2 tk = a.wg_task[blockIdx.x];
    pl = tk.x; rb = tk.y;
    if (pl == 0xFFFFFFFFu) return;  // padding of the XCD interleave
  }
  const uint32_t p = pair0 + pl;
  const uint32_t x = a.pair_x[p], y = a.pair_y[p];
  const uint32_t L1 = a.in.len[x], L2 = a.in.len[y];
  const uint32_t row0 = rb * PCT_ROWS_PER_WG;
  if (row0 >= L1) return;
  // dafs.cpp:280-288 and the two sparse matrices every z contributes
  for (uint32_t z = tid; z < N; z += nt) {
    float w = a.sim[(size_t)z * N + x] * a.sim[(size_t)z * N + y];
    if (a.w_pct < 0.0) w *= 1.0 / N;
    else if (z == x || z == y) w *= (1.0 - a.w_pct) / 2;
    else w *= a.w_pct / (N - 2);
    wz[z] = w;
    if (z == x) { a_rp[z] = (uint64_t)a.in.ident_rp; a_ent[z] = (uint64_t)a.in.ident2; }  // mp[x][x]
    else {
      const uint32_t lo = x < z ? x : z, hi = x < z ? z : x;
      const uint32_t tk = a.in.task_of_pair[pair_id(lo, hi, N)];
      const bool fwd = x < z;
      a_rp[z] = (uint64_t)(a.in.rowptr_pool + (a.in.rp_off[tk] + (fwd ? 0 : a.in.len[lo] + 1)));
      a_ent[z] = (uint64_t)(a.in.ent2 + (a.in.pair_off[tk] + (fwd ? 0 : a.in.pair_nnz[tk])));
    }
    if (z == y) { b_rp[z] = (uint64_t)a.in.ident_rp; b_ent[z] = (uint64_t)a.in.ident2; }  // mp[y][y]
    else {
      const uint32_t lo = z < y ? z : y, hi = z < y ? y : z;
      const uint32_t tk = a.in.task_of_pair[pair_id(lo, hi, N)];
      const bool fwd = z < y;
      b_rp[z] = (uint64_t)(a.in.rowptr_pool + (a.in.rp_off[tk] + (fwd ? 0 : a.in.len[lo] + 1)));
      b_ent[z] = (uint64_t)(a.in.ent2 + (a.in.pair_off[tk] + (fwd ? 0 : a.in.pair_nnz[tk])));
    }
  }
  __syncthreads();
  if (rb == 0 && tid == 0) {
    float sw = 0.0f;
    for (uint32_t z = 0; z < N; ++z) sw += wz[z];
    a.sum_w[pl] = sw;
  }
  float* tile = a.tile + a.tile_off[pl];

  const uint32_t i = row0 + (uint32_t)gw;
  const bool rowact = i < L1;
  const uint32_t t8 = t * 8u, zero = 0u, trash = row_cap + t;
  for (uint32_t j = t; j < L2; j += PCT_G) acc[j] = 0.0f;

  // One item = one (z, k) of the row: lane tl - w0 of the group prepares item tl of the chunk -- the a-entry (k, p_ik), then
  // the row pointers of mp[z][y][k] -- in three branch-free pieces, so that round r + 1's a-entry load is in flight while
  // round r's entries are fetched, and its row-pointer load while round r is applied.  Lanes without an item prepare item 0
  // of the chunk again (valid addresses) and get length 0.
  struct half_item { uint32_t z; bool have; pct_u2 ak; };             // ak: the a-entry {k, p_ik}
  struct item_loads { uint32_t z; float pik; bool have; pct_u2 rp; }; // rp: {first entry, end} of the b-row
  for (uint32_t zc = 0; zc < N; zc += PCT_G) {
    auto item_a = [&](uint32_t tl, uint32_t T1) {
      half_item h;
      h.have = tl < T1;
      const uint32_t tc = h.have ? tl : 0u;
      uint32_t zz = (tc >= zoff[8]) ? 8u : 0u;  // the last z-lane whose first item is not behind tl (empty rows share their offset with the next)
      zz += (tc >= zoff[zz + 4]) ? 4u : 0u;
      zz += (tc >= zoff[zz + 2]) ? 2u : 0u;
      zz += (tc >= zoff[zz + 1]) ? 1u : 0u;
      h.z = min(zc + zz, N - 1);
      h.ak = *(const pct_gent*)(zabase[zz] + (uint64_t)tc * 8u);
      return h;
    };
    auto item_b_loads = [&](const half_item& h) {
      item_loads q;
      q.z = h.z; q.have = h.have; q.pik = __uint_as_float(h.ak.y);
      q.rp = *(const pct_gent*)(b_rp[h.z] + (uint64_t)h.ak.x * 4u);    // 4-byte aligned 8-byte load: row pointers k and k + 1
      return q;
    };
    auto item_b_finish = [&](const item_loads& q) {
      pct_item it;
      const uint64_t adr = b_ent[q.z] + (uint64_t)q.rp.x * 8u;
      it.n = q.have ? q.rp.y - q.rp.x : 0u;
      it.pik = q.pik;
      it.w = wz[q.z];
      it.ptr_lo = (uint32_t)adr; it.ptr_hi = (uint32_t)(adr >> 32);
      return it;
    };
    // ---- a-rows of the chunk, one lane per z
    uint32_t na = 0, beg = 0;
    uint64_t entbase = (uint64_t)a.in.ident2;
    {
      const uint32_t z = zc + t;
      if (rowact && z < N) {
        const pct_u2 be = *(const pct_gent*)(a_rp[z] + (uint64_t)i * 4u);
        beg = be.x; na = be.y - be.x;
        entbase = a_ent[z];
      }
    }
    uint32_t incl = na;
#pragma unroll
    for (int o = 1; o < PCT_G; o <<= 1) {
      const uint32_t up = __shfl_up(incl, o, PCT_G);
      if ((int)t >= o) incl += up;
    }
    zoff[t + 1] = incl;
    if (t == 0) zoff[0] = 0;
    zabase[t] = entbase + ((uint64_t)beg - (uint64_t)(incl - na)) * 8u;  // so that item tl of the chunk is at zabase + 8 tl
    const uint32_t T1 = row_bcast<15>(incl);  // items of this row in this chunk
    uint32_t T1max = T1;
#pragma unroll
    for (int o = 16; o < 64; o <<= 1) T1max = max(T1max, (uint32_t)__shfl_xor((int)T1max, o));
    T1max = __builtin_amdgcn_readfirstlane(T1max);
    wave_lds_fence();
    // ---- rounds of 16 items per row, the next round's items prepared while this round runs
    pct_item it = item_b_finish(item_b_loads(item_a(t, T1)));
    for (uint32_t w0 = 0; w0 < T1max; w0 += PCT_G) {
      const half_item hn = item_a(w0 + PCT_G + t, T1);   // a-entry load of the next round: in flight during this round's fetch
      pct_u2 cv[16];
      pct_step<0>::fetch(it, t8, zero, trash, cv);
      const item_loads qn = item_b_loads(hn);            // its row-pointer load: in flight during this round's apply
      pct_step<0>::apply(it, t, trash, cv, acc, pct_long_slots(it.n));
      it = item_b_finish(qn);
    }
    wave_lds_fence();
  }
  if (rowact)
    for (uint32_t j = t; j < L2; j += PCT_G) tile[(size_t)i * L2 + j] = acc[j];
}

__global__ __launch_bounds__(256) void k_pct_emit(pct_match_args a, uint32_t pair0) {
  extern __shared__ uint32_t s_ptrs[];  // rowptr[max_len + 2], colptr[max_len + 2]
  __shared__ uint32_t s_part[256];
  __shared__ unsigned long long s_off;
  __shared__ int s_ok;
  const uint32_t tid = threadIdx.x, nt = blockDim.x;
  const int wave = (int)(tid >> 6), lane = (int)(tid & 63);
  uint32_t* rowptr = s_ptrs;
  uint32_t* colptr = s_ptrs + a.max_len + 2;
  const uint32_t p = pair0 + blockIdx.x;
  const uint32_t x = a.pair_x[p], y = a.pair_y[p];
  const uint32_t L1 = a.in.len[x], L2 = a.in.len[y];
  const float* tile = a.tile + a.tile_off[blockIdx.x];
  const float sum_w = a.sum_w[blockIdx.x];
  // counts: rows by wavefronts, columns by threads
  for (uint32_t i = (uint32_t)wave; i < L1; i += 4) {
    uint32_t c = 0;
    for (uint32_t j0 = 0; j0 < L2; j0 += 64) {
      const uint32_t j = j0 + lane;
      const bool keep = j < L2 && tile[(size_t)i * L2 + j] / sum_w > PCT_CUTOFF;
      c += (uint32_t)__popcll(__ballot(keep));
    }
    if (lane == 0) rowptr[i + 1] = c;
  }
  for (uint32_t j = tid; j < L2; j += nt) {
    uint32_t c = 0;
    for (uint32_t i = 0; i < L1; ++i) c += (tile[(size_t)i * L2 + j] / sum_w > PCT_CUTOFF) ? 1 : 0;
    colptr[j + 1] = c;
  }
  if (tid == 0) { rowptr[0] = 0; colptr[0] = 0; }
  __syncthreads();
  // inclusive prefix sums in LDS (two arrays, chunked over the threads)
  for (int which = 0; which < 2; ++which) {
    uint32_t* arr = (which ? colptr : rowptr) + 1;
    const uint32_t n = which ? L2 : L1;
    const uint32_t chunk = (n + nt - 1) / nt;
    const uint32_t b = tid * chunk < n ? tid * chunk : n, e = b + chunk < n ? b + chunk : n;
    uint32_t sum = 0;
    for (uint32_t k = b; k < e; ++k) sum += arr[k];
    s_part[tid] = sum;
    __syncthreads();
    if (tid == 0) {
      uint32_t run = 0;
      for (uint32_t k = 0; k < nt; ++k) { const uint32_t v = s_part[k]; s_part[k] = run; run += v; }
    }
    __syncthreads();
    uint32_t run = s_part[tid];
    for (uint32_t k = b; k < e; ++k) { run += arr[k]; arr[k] = run; }
    __syncthreads();
  }
  const uint32_t nnz = rowptr[L1];
  if (tid == 0) {
    const unsigned long long o = atomicAdd(a.pool_top, 2ull * nnz);
    s_off = o;
    s_ok = (o + 2ull * nnz <= a.pool_cap) ? 1 : 0;
    if (!s_ok) atomicExch(a.status, DAFS_HIP_EOVERFLOW);
    a.pair_off[p] = o;
    a.pair_nnz[p] = nnz;
  }
  __syncthreads();
  const unsigned long long off = s_off;
  const uint64_t rp = a.rp_off[p];
  for (uint32_t i = tid; i <= L1; i += nt) a.rowptr_pool[rp + i] = rowptr[i];
  for (uint32_t j = tid; j <= L2; j += nt) a.rowptr_pool[rp + L1 + 1 + j] = colptr[j];
  if (!s_ok) return;
  for (uint32_t i = (uint32_t)wave; i < L1; i += 4) {
    unsigned long long pos = off + rowptr[i];
    for (uint32_t j0 = 0; j0 < L2; j0 += 64) {
      const uint32_t j = j0 + lane;
      const float v = j < L2 ? tile[(size_t)i * L2 + j] / sum_w : 0.0f;
      const bool keep = j < L2 && v > PCT_CUTOFF;
      const unsigned long long m = __ballot(keep);
      if (keep) {
        const unsigned long long q = pos + (uint32_t)__popcll(m & ((1ull << lane) - 1));
        a.col[q] = j; a.val[q] = v;
      }
      pos += (uint32_t)__popcll(m);
    }
  }
  for (uint32_t j = tid; j < L2; j += nt) {
    unsigned long long pos = off + nnz + colptr[j];
    for (uint32_t i = 0; i < L1; ++i) {
      const float v = tile[(size_t)i * L2 + j] / sum_w;
      if (v > PCT_CUTOFF) { a.col[pos] = i; a.val[pos] = v; ++pos; }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// relax_basepairing_probability (dafs.cpp:326-375), same machinery.  Row i of sequence x receives
//   bp[y][k][l] * mp[x][y][i][k] * mp[y][x][l][j] * w_y   for i < j, in the order y, k, l, j.
// A group first fetches the rows mp[x][y][i] of 16 sequences y; every (y, k) then opens the row bp[y][k],
// whose entries (l, p_kl) are the "items": each one adds the row mp[y][x][l], scaled, to the accumulator.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pct_bp_rows(pct_bp_args a, uint32_t row_cap) {
  extern __shared__ unsigned char s_raw[];
  const uint32_t N = a.mp.nseq;
  const uint32_t tid = threadIdx.x, nt = blockDim.x;
  const int t = (int)(tid & 15), gw = (int)(tid >> 4);
  // per y: bases of mp[x][y] (a-side) and of mp[y][x] (c-side), and w_y
  uint64_t* a_rp = (uint64_t*)s_raw;
  uint64_t* a_ent = a_rp + N;
  uint64_t* c_rp = a_ent + N;
  uint64_t* c_ent = c_rp + N;
  float* wy = (float*)(c_ent + N);
  unsigned char* gbase = (unsigned char*)(wy + ((N + 1) & ~1u)) + (size_t)gw * ((size_t)row_cap * 4 + 16 * 8 + 20 * 4 + 16 * 4);
  uint64_t* zaent = (uint64_t*)gbase;
  uint32_t* zoff = (uint32_t*)(zaent + 16);
  uint32_t* zid = zoff + 20;
  float* acc = (float*)(zid + 16);

  const uint32_t x = blockIdx.x;
  const uint32_t L1 = a.mp.len[x];
  const uint32_t row0 = blockIdx.y * PCT_ROWS_PER_WG;
  if (row0 >= L1) return;
  for (uint32_t y = tid; y < N; y += nt) {
    float w = a.sim[(size_t)y * N + x];  // dafs.cpp:341-348
    if (a.w_pct < 0.0) w *= 1.0 / N;
    else if (y == x) w *= 1.0 - a.w_pct;
    else w *= a.w_pct / (N - 1);
    wy[y] = w;
    a_rp[y] = 0; a_ent[y] = 0; c_rp[y] = 0; c_ent[y] = 0;
    if (y != x) {
      const uint32_t lo = x < y ? x : y, hi = x < y ? y : x;
      const uint32_t tk = a.mp.task_of_pair[pair_id(lo, hi, N)];
      const uint64_t rp = a.mp.rp_off[tk], ent = a.mp.pair_off[tk], nz = a.mp.pair_nnz[tk];
      const uint64_t rp2 = rp + a.mp.len[lo] + 1, ent2 = ent + nz;  // the transposed half of the task
      a_rp[y] = x < y ? rp : rp2; a_ent[y] = x < y ? ent : ent2;    // mp[x][y]
      c_rp[y] = x < y ? rp2 : rp; c_ent[y] = x < y ? ent2 : ent;    // mp[y][x]
    }
  }
  __syncthreads();
  if (blockIdx.y == 0 && tid == 0) {
    float sw = 0.0f;
    for (uint32_t y = 0; y < N; ++y) sw += wy[y];
    a.sum_w[x] = sw;
  }
  float* tile = a.tile + a.tile_off[x];

  const uint32_t i = row0 + (uint32_t)gw;
  const bool rowact = i < L1;
  for (uint32_t j = (uint32_t)t; j < L1; j += PCT_G) acc[j] = 0.0f;
  for (uint32_t yc = 0; yc < N; yc += PCT_G) {
    // ---- rows mp[x][y][i] of the chunk, one lane per y (y == x: the identity row {(i, 1)})
    uint32_t na = 0;
    {
      const uint32_t y = yc + (uint32_t)t;
      uint64_t ent = 0;
      if (rowact && y < N) {
        if (y == x) na = 1;
        else {
          const uint32_t beg = a.mp.rowptr_pool[a_rp[y] + i], end = a.mp.rowptr_pool[a_rp[y] + i + 1];
          na = end - beg;
          ent = a_ent[y] + beg;
        }
      }
      zaent[t] = ent;
      zid[t] = y;
    }
    uint32_t incl = na;
#pragma unroll
    for (int o = 1; o < PCT_G; o <<= 1) {
      const uint32_t up = __shfl_up(incl, o, PCT_G);
      if (t >= o) incl += up;
    }
    zoff[t + 1] = incl;
    if (t == 0) zoff[0] = 0;
    const uint32_t T1 = row_bcast<15>(incl);
    uint32_t T1max = T1;
#pragma unroll
    for (int o = 16; o < 64; o <<= 1) T1max = max(T1max, (uint32_t)__shfl_xor((int)T1max, o));
    T1max = __builtin_amdgcn_readfirstlane(T1max);
    wave_lds_fence();
    // ---- rounds of 16 (y, k): each lane opens bp[y][k]
    for (uint32_t w0 = 0; w0 < T1max; w0 += PCT_G) {
      const uint32_t tl = w0 + (uint32_t)t;
      uint32_t my_y = 0, my_nb = 0, my_lo = 0, my_hi = 0;
      float my_pik = 0.0f;
      if (tl < T1) {
        uint32_t zz = 0;
#pragma unroll
        for (int q = 1; q < PCT_G; ++q) zz += (tl >= zoff[q]) ? 1u : 0u;
        my_y = zid[zz];
        uint32_t k = i;
        my_pik = 1.0f;
        if (my_y != x) { const uint2 cv = a.mp.ent2[zaent[zz] + (tl - zoff[zz])]; k = cv.x; my_pik = __uint_as_float(cv.y); }
        const row_ref b = bp_row(a.bp, my_y, k);  // (l, p_kl)
        my_nb = b.n;
        const uint64_t off = (uint64_t)(b.col - a.bp.col);
        my_lo = (uint32_t)off; my_hi = (uint32_t)(off >> 32);
      }
      // ---- the 16 (y, k) of the round in turn: the entries of bp[y][k] are the items
      for (int q = 0; q < PCT_G; ++q) {
        const uint32_t nb = (uint32_t)__shfl((int)my_nb, q, PCT_G);
        uint32_t nbmax = nb;
#pragma unroll
        for (int o = 16; o < 64; o <<= 1) nbmax = max(nbmax, (uint32_t)__shfl_xor((int)nbmax, o));
        nbmax = __builtin_amdgcn_readfirstlane(nbmax);
        if (nbmax == 0) continue;
        const uint32_t yq = (uint32_t)__shfl((int)my_y, q, PCT_G);
        const float pik = __shfl(my_pik, q, PCT_G);
        const uint64_t boff = ((uint64_t)(uint32_t)__shfl((int)my_hi, q, PCT_G) << 32) | (uint32_t)__shfl((int)my_lo, q, PCT_G);
        for (uint32_t e0 = 0; e0 < nbmax; e0 += PCT_G) {
          pct_item it;
          it.ptr_lo = 0; it.ptr_hi = 0; it.n = 0; it.pik = 0.0f; it.w = 0.0f;
          if (e0 + (uint32_t)t < nb) {
            const uint32_t l = a.bp.col[boff + e0 + (uint32_t)t];
            const float pkl = a.bp.val[boff + e0 + (uint32_t)t];
            it.pik = pkl * pik;  // p_kl * p_ik, then * p_jl * w (:359, :368)
            it.w = wy[yq];
            if (yq == x) {  // mp[x][x][l] = {(l, 1)}
              const uint64_t adr = (uint64_t)(a.mp.ident2 + l);
              it.n = 1; it.ptr_lo = (uint32_t)adr; it.ptr_hi = (uint32_t)(adr >> 32);
            } else {
              const uint32_t cb = a.mp.rowptr_pool[c_rp[yq] + l], ce = a.mp.rowptr_pool[c_rp[yq] + l + 1];
              const uint64_t ptr = c_ent[yq] + cb;
              const uint64_t adr = (uint64_t)(a.mp.ent2 + ptr);
              it.n = ce - cb; it.ptr_lo = (uint32_t)adr; it.ptr_hi = (uint32_t)(adr >> 32);
            }
          }
          uint32_t jc[16];
          float pv[16];
          pct_round<0>::fetch(a.mp.ent2, it, t, jc, pv);
          pct_round<0>::apply(a.mp.ent2, it, t, i + 1, jc, pv, acc, pct_long_slots(it.n));
        }
      }
    }
    wave_lds_fence();
  }
  if (rowact)
    for (uint32_t j = (uint32_t)t; j < L1; j += PCT_G) tile[(size_t)i * L1 + j] = acc[j];
}

// thresholded upper-triangular CSR of one sequence's tile
__global__ __launch_bounds__(256) void k_pct_bp_emit(pct_bp_args a) {
  extern __shared__ uint32_t s_ptrs[];  // rowptr[max_len + 2]
  __shared__ uint32_t s_part[256];
  __shared__ unsigned long long s_off;
  __shared__ int s_ok;
  const uint32_t tid = threadIdx.x, nt = blockDim.x;
  const int wave = (int)(tid >> 6), lane = (int)(tid & 63);
  uint32_t* rowptr = s_ptrs;
  const uint32_t x = blockIdx.x;
  const uint32_t L1 = a.mp.len[x];
  const float* tile = a.tile + a.tile_off[x];
  const float sum_w = a.sum_w[x];
  for (uint32_t i = (uint32_t)wave; i < L1; i += 4) {
    uint32_t c = 0;
    for (uint32_t j0 = 0; j0 < L1; j0 += 64) {
      const uint32_t j = j0 + lane;
      const bool keep = j < L1 && j > i && tile[(size_t)i * L1 + j] / sum_w > PCT_CUTOFF;
      c += (uint32_t)__popcll(__ballot(keep));
    }
    if (lane == 0) rowptr[i + 1] = c;
  }
  if (tid == 0) rowptr[0] = 0;
  __syncthreads();
  {
    uint32_t* arr = rowptr + 1;
    const uint32_t n = L1;
    const uint32_t chunk = (n + nt - 1) / nt;
    const uint32_t b = tid * chunk < n ? tid * chunk : n, e = b + chunk < n ? b + chunk : n;
    uint32_t sum = 0;
    for (uint32_t k = b; k < e; ++k) sum += arr[k];
    s_part[tid] = sum;
    __syncthreads();
    if (tid == 0) {
      uint32_t run = 0;
      for (uint32_t k = 0; k < nt; ++k) { const uint32_t v = s_part[k]; s_part[k] = run; run += v; }
    }
    __syncthreads();
    uint32_t run = s_part[tid];
    for (uint32_t k = b; k < e; ++k) { run += arr[k]; arr[k] = run; }
    __syncthreads();
  }
  if (tid == 0) {
    const uint32_t n = rowptr[L1];
    const unsigned long long o = atomicAdd(a.pool_top, (unsigned long long)n);
    s_off = o;
    s_ok = (o + n <= a.pool_cap) ? 1 : 0;
    if (!s_ok) atomicExch(a.status, DAFS_HIP_EOVERFLOW);
    a.out_off[x] = o;
    a.out_nnz[x] = n;
  }
  __syncthreads();
  const unsigned long long off = s_off;
  const uint64_t rp = a.bp.rp_off[x];  // same row-pointer layout as the input store
  for (uint32_t i = tid; i <= L1; i += nt) a.out_rowptr[rp + i] = rowptr[i];
  if (!s_ok) return;
  for (uint32_t i = (uint32_t)wave; i < L1; i += 4) {
    unsigned long long pos = off + rowptr[i];
    for (uint32_t j0 = 0; j0 < L1; j0 += 64) {
      const uint32_t j = j0 + lane;
      const float v = j < L1 ? tile[(size_t)i * L1 + j] / sum_w : 0.0f;
      const bool keep = j < L1 && j > i && v > PCT_CUTOFF;
      const unsigned long long m = __ballot(keep);
      if (keep) {
        const unsigned long long q = pos + (uint32_t)__popcll(m & ((1ull << lane) - 1));
        a.out_col[q] = j; a.out_val[q] = v;
      }
      pos += (uint32_t)__popcll(m);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// calculate_similarity_score (dafs.cpp:713-764) from stored rows, for matching probabilities that
// were supplied rather than computed (dafs_hip_set_mp): one thread per pair, the previous DP row
// of all pairs interleaved in scratch so that the threads of a wavefront touch neighbouring words.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_mp_sim(mp_store_dev in, const uint32_t* pair_x, const uint32_t* pair_y, uint32_t npairs, float* task_sim,
                                                float* row_dp, int* row_tr) {
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= npairs) return;
  const uint32_t x = pair_x[p], y = pair_y[p];
  const uint32_t L1 = in.len[x], L2 = in.len[y];
  for (uint32_t j = 0; j <= L2; ++j) { row_dp[(size_t)j * npairs + p] = 0.0f; row_tr[(size_t)j * npairs + p] = 0; }
  float dp = 0.0f;
  int tr = 0;
  for (uint32_t i = 1; i <= L1; ++i) {
    const row_ref r = mp_row(in, x, y, i - 1);
    uint32_t e = 0;
    float ddp = 0.0f, ldp = 0.0f;  // (i-1, j-1) and (i, j-1); column 0 is the border
    int dtr = 0, ltr = 0;
    for (uint32_t j = 1; j <= L2; ++j) {
      const float udp = row_dp[(size_t)j * npairs + p];
      const int utr = row_tr[(size_t)j * npairs + p];
      const bool entry = e < r.n && r.col[e] == j - 1;
      if (entry) {
        dp = ddp + r.val[e]; tr = dtr + 1;
        if (dp < ldp) { dp = ldp; tr = ltr + 1; }
        if (dp < udp) { dp = udp; tr = utr + 1; }
        ++e;
      } else {
        dp = ldp; tr = ltr + 1;
        if (dp < udp) { dp = udp; tr = utr + 1; }
      }
      ddp = udp; dtr = utr;
      row_dp[(size_t)j * npairs + p] = dp;
      row_tr[(size_t)j * npairs + p] = tr;
      ldp = dp; ltr = tr;
    }
  }
  task_sim[p] = dp / (float)tr;  // dafs.cpp:763
}

int mp_sim_launch(mp_store_dev in, const uint32_t* pair_x, const uint32_t* pair_y, uint32_t npairs, float* task_sim, float* row_dp, int* row_tr,
                  hipStream_t st) {
  if (!npairs) return DAFS_HIP_OK;
  STAGE_LAUNCH(ST_MP_SIM, st) hipLaunchKernelGGL(k_mp_sim, dim3((npairs + 255) / 256), dim3(256), 0, st, in, pair_x, pair_y, npairs, task_sim, row_dp, row_tr);
  return hip_check(hipGetLastError()) ? DAFS_HIP_ELAUNCH : DAFS_HIP_OK;
}

static const size_t kPctLdsBytes = 150 * 1024;  // leave room for the static LDS and alignment

size_t pct_rows_lds_bytes(uint32_t nseq, uint32_t row_cap) {
  return (size_t)nseq * 32 + (((size_t)nseq + 1) & ~(size_t)1) * 4 + (size_t)PCT_ROWS_PER_WG * ((size_t)row_cap * 4 + 16 * 8 + 20 * 4 + 16 * 4) + 64;
}
// k_pct_rows (round 3 layout: trash cells behind every accumulator row)
size_t pct_rows2_lds_bytes(uint32_t nseq, uint32_t row_cap) {
  return (size_t)nseq * 32 + (((size_t)nseq + 1) & ~(size_t)1) * 4 + (size_t)PCT_ROWS_PER_WG * pct_group_words(row_cap) * 4 + 64;
}

// pairs [pair0, pair0 + count) whose tiles (a.tile, a.tile_off, a.sum_w) have been laid out by the caller
// ---------------------------------------------------------------------------------------------
// DAFS::relax_fourway_consistency, reference src/dafs.cpp:377-444 (option -f): every entry (a, b) of mp[x][y] becomes
//   p_ab (1 - w)  +  w * sum over base pairs (a, j) of x, (b, l) of y with (j, l) in mp[x][y] of  p_aj p_bl p_jl
//                 +  w * sum over base pairs (i, a) of x, (k, b) of y with (i, k) in mp[x][y] of  p_ia p_kb p_ik
// The reference scatters the last sum from the rows i < a while it walks them, so cell (a, b) receives it first, in
// the order (i ascending, k ascending), then its own base term (a double product added to the float cell), then the
// middle sum in the order (j ascending, l ascending) -- the order a thread per entry reproduces here by gathering.
// Cells that are not entries of mp[x][y] stay 0.  One workgroup per pair writes the dense L1 x L2 tile that k_pct_emit
// turns into rows, transposed rows and counts (its division by sum_w = 1 is exact).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_fourway_rows(pct_match_args a, uint32_t pair0) {
  const uint32_t tid = threadIdx.x, nt = blockDim.x;
  const uint32_t p = pair0 + blockIdx.x;
  const uint32_t x = a.pair_x[p], y = a.pair_y[p];
  const uint32_t L1 = a.in.len[x], L2 = a.in.len[y];
  float* tile = a.tile + a.tile_off[blockIdx.x];
  for (size_t c = tid; c < (size_t)L1 * L2; c += nt) tile[c] = 0.0f;
  if (tid == 0) a.sum_w[blockIdx.x] = 1.0f;
  __syncthreads();
  const float w = a.w_f;
  const uint32_t t = a.in.task_of_pair[pair_id(x, y, a.in.nseq)];
  const uint32_t* rp = a.in.rowptr_pool + a.in.rp_off[t];  // row pointers of mp[x][y]
  const uint32_t nnz = a.in.pair_nnz[t];
  const uint32_t* mcol = a.in.col + a.in.pair_off[t];
  const float* mval = a.in.val + a.in.pair_off[t];
  for (uint32_t e = tid; e < nnz; e += nt) {
    uint32_t lo = 0, hi = L1;  // row of entry e: the largest r with rp[r] <= e
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (rp[mid] <= e) lo = mid; else hi = mid; }
    const uint32_t ra = lo, cb = mcol[e];
    const float p_ab = mval[e];
    float post = 0.0f;
    // what the rows i < a delivered: base pairs (i, a) of x in the order of i, then the entries (i, k) of that row
    for (uint32_t i = 0; i < ra; ++i) {
      const row_ref bi = bp_row(a.bp, x, i);
      float p_ia = 0.0f;
      bool has = false;
      for (uint32_t q = 0; q < bi.n; ++q)
        if (bi.col[q] == ra) { p_ia = bi.val[q]; has = true; }
      if (!has) continue;
      for (uint32_t q = rp[i]; q < rp[i + 1]; ++q) {
        const uint32_t k = mcol[q];
        const float p_ik = mval[q];
        const row_ref bk = bp_row(a.bp, y, k);
        for (uint32_t r = 0; r < bk.n; ++r)
          if (bk.col[r] == cb) post += p_ia * bk.val[r] * p_ik * w;  // :414
      }
    }
    post = (float)((double)post + (double)p_ab * (1.0 - (double)w));  // :397
    {
      const row_ref ba = bp_row(a.bp, x, ra);
      const row_ref bb = bp_row(a.bp, y, cb);
      for (uint32_t q = 0; q < ba.n; ++q) {
        const uint32_t j = ba.col[q];
        const float p_aj = ba.val[q];
        uint32_t l1 = rp[j], e1 = rp[j + 1], l2 = 0;
        while (l1 != e1 && l2 != bb.n) {  // :402-419
          const uint32_t c1 = mcol[l1], c2 = bb.col[l2];
          if (c1 < c2) ++l1;
          else if (c1 > c2) ++l2;
          else { post += p_aj * bb.val[l2] * mval[l1] * w; ++l1; ++l2; }  // :413
        }
      }
    }
    tile[(size_t)ra * L2 + cb] = post;
  }
}

__global__ __launch_bounds__(256) void k_mp_interleave(const uint32_t* __restrict__ col, const float* __restrict__ val, uint2* __restrict__ ent2, uint64_t n) {
  for (uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (uint64_t)gridDim.x * blockDim.x)
    ent2[e] = make_uint2(col[e], __float_as_uint(val[e]));
}

__global__ __launch_bounds__(256) void k_mp_ident(uint2* __restrict__ ident2, uint32_t* __restrict__ ident_rp, uint32_t n) {
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < n) { ident2[k] = make_uint2(k, 0x3F800000u); ident_rp[k] = k; }
}
int pct_ident_launch(uint2* ident2, uint32_t* ident_rp, uint32_t n, hipStream_t st) {
  if (!n) return DAFS_HIP_OK;
  hipLaunchKernelGGL(k_mp_ident, dim3((n + 255) / 256), dim3(256), 0, st, ident2, ident_rp, n);
  return hip_check(hipGetLastError()) ? DAFS_HIP_ELAUNCH : DAFS_HIP_OK;
}

int pct_interleave_launch(const uint32_t* col, const float* val, uint2* ent2, uint64_t n, hipStream_t st) {
  if (!n) return DAFS_HIP_OK;
  const uint64_t blocks = (n + 255) / 256;
  STAGE_LAUNCH(ST_MP_INTERLEAVE, st) hipLaunchKernelGGL(k_mp_interleave, dim3((uint32_t)(blocks < 65536 ? blocks : 65536)), dim3(256), 0, st, col, val, ent2, n);
  return hip_check(hipGetLastError()) ? DAFS_HIP_ELAUNCH : DAFS_HIP_OK;
}

int pct_fourway_launch(pct_match_args a, uint32_t max_len, uint32_t pair0, uint32_t count, hipStream_t st) {
  if (!count) return DAFS_HIP_OK;
  a.max_len = max_len;
  STAGE_LAUNCH(ST_FOURWAY_ROWS, st) hipLaunchKernelGGL(k_fourway_rows, dim3(count), dim3(256), 0, st, a, pair0);
  if (hip_check(hipGetLastError())) return DAFS_HIP_ELAUNCH;
  STAGE_LAUNCH(ST_PCT_EMIT, st) hipLaunchKernelGGL(k_pct_emit, dim3(count), dim3(256), (size_t)2 * (max_len + 2) * 4, st, a, pair0);
  return hip_check(hipGetLastError()) ? DAFS_HIP_ELAUNCH : DAFS_HIP_OK;
}

int pct_match_launch(pct_match_args a, uint32_t max_len, uint32_t pair0, uint32_t count, hipStream_t st) {
  if (!count) return DAFS_HIP_OK;
  // The four rows a wavefront accumulates sit pct_group_words(row_cap) = row_cap + 84 words apart in LDS, and the two rows
  // of a 32-lane half add into nearly the same 16-column window (rows i and i+1 of one pair): row_cap = 28 (mod 32) puts the
  // second row sixteen banks away from the first (four banks away: two lanes per bank on almost every update -- 1.9 G
  // conflict cycles per 2.3 G LDS instructions at N = 128).
  const uint32_t row_cap = max_len + ((28u - max_len) & 31u);
  const size_t lds = pct_rows2_lds_bytes(a.in.nseq, row_cap);
  if (lds > kPctLdsBytes) return DAFS_HIP_ETOOLONG;
  a.max_len = max_len;
  static bool attr[16] = {false};
  if (!lds_optin_once((const void*)k_pct_rows, (int)kPctLdsBytes, attr)) return DAFS_HIP_ELAUNCH;
  if (a.wg_task) {
    STAGE_LAUNCH(ST_PCT_ROWS, st) hipLaunchKernelGGL(k_pct_rows, dim3(a.wg_tasks), dim3(256), lds, st, a, pair0, row_cap);
  } else {
    STAGE_LAUNCH(ST_PCT_ROWS, st) hipLaunchKernelGGL(k_pct_rows, dim3(count, (max_len + PCT_ROWS_PER_WG - 1) / PCT_ROWS_PER_WG), dim3(256), lds, st, a, pair0, row_cap);
  }
  if (hip_check(hipGetLastError())) return DAFS_HIP_ELAUNCH;
  STAGE_LAUNCH(ST_PCT_EMIT, st) hipLaunchKernelGGL(k_pct_emit, dim3(count), dim3(256), (size_t)2 * (max_len + 2) * 4, st, a, pair0);
  return hip_check(hipGetLastError()) ? DAFS_HIP_ELAUNCH : DAFS_HIP_OK;
}

int pct_bp_launch(pct_bp_args a, uint32_t max_len, hipStream_t st) {
  // The four rows a wavefront accumulates sit row_cap + 68 words apart in LDS, and the two rows of a 32-lane half
  // add into nearly the same 16-column window (rows i and i+1 of one pair): row_cap = 12 (mod 32) puts the second row
  // sixteen banks away from the first (a multiple of 16, as before, put it four banks away: two lanes per bank on
  // almost every update -- 1.9 G conflict cycles per 2.3 G LDS instructions at N = 128).
  const uint32_t row_cap = max_len + ((12u - max_len) & 31u);
  const size_t lds = pct_rows_lds_bytes(a.mp.nseq, row_cap);
  if (lds > kPctLdsBytes) return DAFS_HIP_ETOOLONG;
  a.max_len = max_len;
  static bool attr[16] = {false};
  if (!lds_optin_once((const void*)k_pct_bp_rows, (int)kPctLdsBytes, attr)) return DAFS_HIP_ELAUNCH;
  STAGE_LAUNCH(ST_PCT_BP_ROWS, st) hipLaunchKernelGGL(k_pct_bp_rows, dim3(a.mp.nseq, (max_len + PCT_ROWS_PER_WG - 1) / PCT_ROWS_PER_WG), dim3(256), lds, st, a, row_cap);
  if (hip_check(hipGetLastError())) return DAFS_HIP_ELAUNCH;
  STAGE_LAUNCH(ST_PCT_BP_EMIT, st) hipLaunchKernelGGL(k_pct_bp_emit, dim3(a.mp.nseq), dim3(256), (size_t)(max_len + 2) * 4, st, a);
  return hip_check(hipGetLastError()) ? DAFS_HIP_ELAUNCH : DAFS_HIP_OK;
}

}  // namespace dafs
