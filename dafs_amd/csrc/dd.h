// dafs_amd/csrc/dd.h -- device-side descriptors of the progressive phase (dd.hip):
// one dd_node per guide-tree node being solved, all pointers into device memory.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "sparse_view.h"

namespace dafs {

struct nuss_ws {      // SparseNussinov work arrays for one problem of size L
  float* dp;          // L*L
  uint32_t* tr;       // L*L
  uint32_t* ck;       // L*L  candidate list of column j at ck[j*L ..], insertion order (k descending)
  float* cv;          // L*L
  uint32_t* cc;       // L    candidates per column
};

struct dd_node {
  uint32_t n1, n2, L1, L2;
  // child alignments: per row the sequence index, per (row, column) the residue rank or NONE,
  // and per (row, residue) its column
  const uint32_t *seq1, *seq2, *rank1, *rank2, *idx1, *idx2, *idxoff1, *idxoff2;
  // averaged posteriors (dafs.cpp:513-607) and Lagrange multipliers
  float *p_x, *p_y, *p_z, *q_x, *q_y, *q_z;
  nuss_ws wx, wy;
  float* nw_edge;     // 2*(L1+2): the last column of a panel of the alignment DP, for the next panel (nw_wave_reg)
  uint8_t* tr_z;      // panels*(L1+1)*512: traceback codes of the alignment DP when they are not packed in LDS -- a 64-bit slot per (panel, row, lane)
  uint32_t nw_w;      // columns per lane of the alignment DP (dd_nw_cols; DAFS_HIP_DD_WIDE=1 makes it 1): second alignments beyond
                      // 64*nw_w - 1 columns run as panels of 64*nw_w columns; also the layout of pz_s / qz_s (nw_idx)
  uint8_t *trb_x, *trb_y;   // L(L+1)/2 each: Nussinov traceback codes 0..4 (HBM copies)
  uint32_t *trk_x, *trk_y;  // L*L each: bifurcation code of the cells whose traceback code is 4
  float *s_x, *s_y;         // (L+63)*ceil(L/64)*64 each: pair scores w*(p-th)-q in sweep order of the folding DP; null when the
                            // folding has no register form (more than DD_WFOLD columns per lane)
  float *s_xs, *s_ys;       // L*Lp each (Lp = L rounded up to 64): the same scores stored by span, S[(j-i)*Lp + i], for the span form
                            // (nuss_wave_span); null when no launch of this node can take that form
  float *pz_s, *qz_s;       // panels*(L1+63)*nw_w*64 each: p_z, q_z in sweep order of the alignment DP, panel by panel
  uint32_t lds_flags;       // LDS plan: bit 0 packed alignment traceback, bit 1 / bit 2 fast form of the x / y folding DP, bit 3 / 4 shared region / codes in HBM,
                            // bit 6 span form of both folding DPs side by side (whole triangles in LDS; scores from s_xs / s_ys)
  uint32_t* env;      // 2*(L1+1)
  uint32_t* env4;     // 2*(L1+130): the same envelope for the register-resident alignment DP -- {max(first,1), second} of row r at
                      // index r + 64, the empty range {1, 0} for the 64 rows before row 1 and the 65 behind row L1 (no clamping, no selects)
  // sparse structure of p_x / p_y / p_z (> CUTOFF) and of the consensus base pairs
  int32_t *xmap, *ymap, *zmap;    // dense cell -> entry id (px / py / cz lists) or -1
  uint32_t *px_ptr, *px_j, *py_ptr, *py_l, *pz_ptr, *pz_k, *cz_ptr, *cz_k;
  uint8_t *cx_flag, *cy_flag, *cz_flag;
  uint32_t* cbp_cnt;              // per px entry
  uint32_t* cbp;                  // 8 per consensus base pair: i j k l pxid pyid zid1 zid2
  uint32_t ncbp_cap;
  int32_t *tx, *ty, *tz;          // per entry violation counters
  float* sw;                      // ncbp: positive s_w, compacted in cbp order
  // results
  uint32_t *x, *y, *z;
  float* score;                   // [1]
  uint32_t* info;                 // [16]: ncbp, iterations (next iteration while paused), violated, status, slow-x, slow-y,
                                  //       started, paused; [8..13] optional phase ticks
  float* fstate;                  // [4]: c, eta, previous dual value of a paused node
  // split mode: the two folding DPs of this node run on workgroups of their own (blockIdx.y = 1, 2) next to
  // the leader (blockIdx.y = 0, alignment DP + constraints + updates); sync[0] go / exit, [1] x done, [2] y done,
  // [3] [4] the folding scores.  fold_fast: bit 0 / 1 the fast form of x / y fits the folder's LDS, bit 2 / 3 the same with the
  // codes in HBM, bit 4 / 5 the span form fits it (preferred), bit 6 / 7 no register form but the workgroup form fits
  // (nuss_wg_span; needs s_xs / s_ys) with K = bits 8..11 / 12..15 candidates per column in LDS.
  uint32_t* sync;
  uint32_t split, fold_fast;
};

struct dd_params {
  float w, eta0, th_a, th_s;
  uint32_t t_max;
  int force_iters;
  int stamps;  // accumulate per-phase timing into info[8..13] (tuning aid)
  int skip_xy; // dafs_dd_params::skip_uncoupled_folds
  int span_one_wave;       // DAFS_HIP_DD_SPAN_MW=0 (tests, tuning): the folders keep the span form on one wavefront
  int debug_lose_folders;  // DAFS_HIP_DD_LOSE_FOLDERS=1 (tests): the leader of a split node treats its folders as lost at once
  uint32_t slice;  // at most this many iterations per launch (0 = run to the end); a node that is cut short is
                   // marked paused and continues from where it stopped at the next launch
  // A second bound on a launch, in time: a node also pauses at the first iteration end past `budget` ticks (100 MHz,
  // wall_clock64) after the reference tick -- its workgroup's own start, or, when t_ref is given, the tick the round's
  // first launch left there (t_ref_write: this launch is the one that leaves it), so that launches started at different
  // moments of a round stop together (dafs_hip_nodes_round).  0 = no time bound.
  unsigned long long budget;
  unsigned long long* t_ref;
  int t_ref_write;
};

int dd_avg_launch(const dd_node* d_nodes, uint32_t nnodes, uint32_t max_len, mp_store_dev mp, bp_store_dev bp, int one_row, int coop, hipStream_t st);
int dd_lists_launch(const dd_node* d_nodes, uint32_t nnodes, uint32_t max_len1, dd_params prm, uint32_t* d_ncbp, hipStream_t st);  // d_ncbp[b]: consensus pairs of node b
int dd_cbp_fill_launch(const dd_node* d_nodes, uint32_t nnodes, uint32_t max_len1, dd_params prm, hipStream_t st);
#define DD_WREG 8     // widest lane (columns) of the register-resident alignment DP, and of the folding DP with its codes in LDS
#define DD_WFOLD 16   // widest lane of the register-resident folding DP (codes in HBM beyond DD_WREG)
#define DD_WNW 16     // widest lane of the register-resident alignment DP with its codes packed in LDS (second alignments up to 1023 columns)
#define DD_WNWG 32    // widest lane of the same DP with its codes in HBM, one 64-bit slot per row and lane (up to 2047 columns)
// columns per lane of the alignment DP, which is also the layout of its sweep-order inputs (nw_skew): beyond DD_WNW the
// register form exists for multiples of four only
static inline __host__ __device__ uint32_t dd_nw_cols(uint32_t L2) {
  const uint32_t W = (L2 + 64) / 64;
  return W > DD_WNWG ? DD_WNWG : (W > DD_WNW ? (W + 3) & ~3u : W);
}
static inline __host__ __device__ uint32_t dd_nw_panels(uint32_t L2, uint32_t W) { return (L2 + 64 * W) / (64 * W); }  // columns 0 .. L2
#define DD_CAP 4  // candidates per column kept in LDS by the fast folding DP
// LDS words of the in-flight rows of a fast folding DP: one row of L values per active lane (the lanes own
// ceil(L/64) columns each, so ceil(L / that) of them are at work).  The previous-row buffers and candidate counters of
// the HBM-table form borrow the same words when that form has to run (the ring is idle then), hence the floor.
// Columns per lane of the folding wave DPs.  Up to 512 columns all 64 lanes are used; from there to 768 only 48, so
// that the rows in flight (one per lane at work, L values each) still fit LDS and the register form can run with up
// to DD_WFOLD columns per lane; wider alignments use 64 lanes again and the HBM-table form.
static inline __host__ __device__ uint32_t dd_fold_cols(uint32_t L) { return (L > 512 && L <= 768) ? (L + 47) / 48 : (L + 63) / 64; }
static inline __host__ __device__ uint32_t dd_ring_rows(uint32_t L) { const uint32_t W = dd_fold_cols(L); return W ? (L + W - 1) / W : 0; }
static inline __host__ __device__ uint32_t dd_slow_words(uint32_t L) { return 2 * dd_fold_cols(L) * 64 + L; }
static inline __host__ __device__ uint32_t dd_ring_words(uint32_t L) {
  const uint32_t a = dd_ring_rows(L) * L, b = dd_slow_words(L);
  return a > b ? a : b;
}
// Span form of a folding DP (dd.hip, nuss_wave_span): packed codes, the whole dp triangle, DD_CAP candidate values and
// row offsets for each of L + 1 columns (16-byte slots), DD_CAP split rows per column.  Each part starts on 16 bytes.
#define DD_SPAN_LMAX 256  // four row slots per lane
static inline __host__ __device__ uint32_t dd_span_nib_words(uint32_t L) { return (uint32_t)((((size_t)L * (L + 1) / 2 + 7) / 8 + 3) & ~(size_t)3); }
static inline __host__ __device__ uint32_t dd_span_tri_words(uint32_t L) { return (uint32_t)(((size_t)L * (L + 1) / 2 + 3) & ~(size_t)3); }
// workgroup form of the folding DP beyond the register forms (nuss_wg_span): three rolling rows of dp values, the candidate
// counters, and the first K candidates (key, value) of every column
static inline __host__ __device__ uint32_t dd_wg_words(uint32_t L, uint32_t K) { return (4 + 2 * K) * ((L + 3) & ~3u); }
static inline __host__ __device__ uint32_t dd_span_words(uint32_t L) { return dd_span_nib_words(L) + dd_span_tri_words(L) + 2 * DD_CAP * (L + 1) + DD_CAP * L; }
// packed traceback table of the alignment DP: two bits per cell, rows padded to whole 32-bit words (16 cells), so that the
// bit position of a lane's cells within their word is the same in every row
static inline __host__ __device__ uint32_t dd_nwtab_row_words(uint32_t L2) { return (L2 + 1 + 15) / 16; }
static inline __host__ __device__ uint32_t dd_nwtab_words(uint32_t L1, uint32_t L2) { return (L1 + 1) * dd_nwtab_row_words(L2); }
static const size_t kDdLdsBudget = 156 * 1024;  // dynamic LDS of k_dd_solve (the CU has 160 KB; ~2.2 KB is static)
int dd_pack_launch(const dd_node* d_nodes, uint32_t nnodes, const uint32_t* d_off, uint32_t* d_out, hipStream_t st);
int dd_solve_launch(const dd_node* d_nodes, uint32_t nnodes, dd_params prm, size_t lds_bytes, bool split, uint32_t* d_paused, hipStream_t st);
// standalone decoders on dense device matrices (one workgroup each)
int nussinov_launch(uint32_t L, const float* p, const float* q, float w, float th, nuss_ws ws, uint32_t* ss, float* score, hipStream_t st);
int nussinov_dense_launch(uint32_t L, const float* p, const float* q, float w, float th, float* dp, uint32_t* tr, uint32_t* stack, uint32_t* ss,
                          float* score, hipStream_t st);
int nw_launch(uint32_t L1, uint32_t L2, const float* p, const float* q, float th, uint32_t* env, int compute_env,
              float* dp, uint8_t* tr, uint32_t* al, float* score, hipStream_t st);

}  // namespace dafs
