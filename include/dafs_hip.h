/* include/dafs_hip.h -- C ABI of libdafs_hip.so, the MI355X (gfx950) implementation of the
 * DAFS probability-matrix + dual-decomposition hot path.
 *
 * Two layers, both plain C (pointers + sizes, no C++/torch types):
 *
 *  L1 "plugin" entry points (dafs_hip_*): host buffers in, host buffers out.  These are what the
 *     reference's four plugin interfaces would bind (reference src/align.h:34-66,
 *     src/fold.h:30-61) -- see INTEGRATION.md for the C++ shim a DAFS maintainer would add.
 *     They own their device workspace and synchronise before returning.
 *
 *  L0 "launch" entry points (dafs_hipk_*): device pointers + a HIP stream, no allocation, no
 *     synchronisation.  Used by L1 and by callers that keep data resident in HBM (bench.py
 *     allocates with torch and passes tensor.data_ptr()).
 *
 * All functions return 0 on success or a negative DAFS_HIP_E* code; no exception crosses the
 * ABI.  dafs_hip_strerror() gives the message the C++ shim throws as `const char*`, the
 * reference's error convention (reference src/dafs.cpp:1893-1910).
 * Indices are uint32; "none" is 0xFFFFFFFF (the reference's -1u, src/nussinov.cpp:267).
 */
#ifndef DAFS_HIP_H
#define DAFS_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DAFS_HIP_NONE 0xFFFFFFFFu

enum {
  DAFS_HIP_OK = 0,
  DAFS_HIP_EINVAL = -1,    /* bad argument (NULL, empty sequence, unknown model) */
  DAFS_HIP_ENODEV = -2,    /* no usable HIP device / HIP runtime error */
  DAFS_HIP_ENOMEM = -3,    /* device or host allocation failed */
  DAFS_HIP_ETOOLONG = -4,  /* sequence longer than the kernels support */
  DAFS_HIP_EOVERFLOW = -5, /* sparse output pool too small (retry with a larger pool) */
  DAFS_HIP_ELAUNCH = -6,   /* kernel launch / execution failure */
  DAFS_HIP_ECOMM = -7      /* the caller's collective (dafs_allgather_fn) reported a failure */
};
const char* dafs_hip_strerror(int code);
/* last HIP runtime error text seen by this thread (diagnostics) */
const char* dafs_hip_last_error(void);

/* Alignment models: reference -a ProbCons | CONTRAlign (src/dafs.cpp:1683-1690) */
enum { DAFS_ALIGN_PROBCONS = 0, DAFS_ALIGN_CONTRALIGN = 1 };
/* Folding models: reference -s CONTRAfold (src/dafs.cpp:1703).  Boltzmann/Vienna need
 * ViennaRNA arithmetic that is not in the reference tree: not provided. */
enum { DAFS_FOLD_CONTRAFOLD = 0 };

/* ------------------------------------------------------------------------------------------
 * L0: pair-HMM posterior kernel (ProbCons 3-state model)
 * Replaces, per pair: PROBCONS::Probcons::ComputePosterior (src/probconsRNA/wrapper.cpp:101-131)
 * + ProbCons::calculate's dense->sparse step (src/align.cpp:60-79) + transpose_mp
 * (src/dafs.cpp:155-167) + calculate_similarity_score (src/dafs.cpp:713-764).
 * ---------------------------------------------------------------------------------------- */

/* One pair job: offsets/lengths of the two residue-code strings inside `codes`. */
typedef struct {
  uint32_t off1, len1; /* sequence x (rows)    */
  uint32_t off2, len2; /* sequence y (columns) */
} dafs_pair_task;

/* Launch geometry chosen by dafs_hipk_pairhmm_plan for a batch. */
typedef struct {
  uint32_t group;        /* lanes cooperating on one pair: 16, 32 or 64           */
  uint32_t width;        /* columns owned by each lane                              */
  uint32_t nwaves;       /* persistent wavefronts launched (4 per workgroup)        */
  uint32_t slab_steps;   /* wavefront steps one slab holds = max(len1)+group        */
  uint64_t scratch_bytes;/* bytes of `scratch` the launch needs                     */
} dafs_pairhmm_plan;

/* 7 residue classes: A C G U T N other ('other' includes the reference's '~' sentinel). */
typedef struct {
  float init[3];     /* log initial distribution M, X, Y   (Defaults.h:19)                  */
  float trans[3][3]; /* log transition [from][to]          (ProbabilisticModel.h:59-79)     */
  float match[7][8]; /* log pair emission, row stride 8    (Defaults.h:30-37)               */
  float ins[8];      /* log single emission                (Defaults.h:26-28)               */
} dafs_pairhmm3_model;

typedef struct {
  const uint8_t* codes;        /* [device] residue class codes 0..6, all sequences concatenated */
  const dafs_pair_task* tasks; /* [device] ntasks jobs, in processing order (longest first)      */
  uint32_t ntasks;
  float th;                    /* keep posterior > th (reference -u, default 0.01)               */
  float* scratch;              /* [device] plan.scratch_bytes                                     */
  uint32_t* queue;             /* [device] one zeroed uint32: dynamic work counter                */
  /* outputs, all [device].  Pair t owns rowptr_pool[rp_off[t] .. +len1+1) (row pointers of
   * mp[x][y], relative to the pair's base) followed by len2+1 row pointers of mp[y][x];
   * entries live at ent_col/ent_val[pair_off[t] .. +nnz) (mp[x][y], rows ascending, columns
   * ascending) and [pair_off[t]+nnz .. +2nnz) (mp[y][x]).  pair_off is assigned by a device-side
   * bump allocator on *pool_top, so pool order is unspecified; contents are deterministic. */
  const uint64_t* rp_off;
  uint32_t* rowptr_pool;
  uint32_t* ent_col;
  float* ent_val;
  unsigned long long* pool_top; /* zeroed before launch */
  uint64_t pool_cap;            /* capacity of ent_col/ent_val in entries */
  uint64_t* pair_off;
  uint32_t* pair_nnz;
  float* sim;                   /* similarity score per task (dafs.cpp:763) */
  int* status;                  /* zeroed; set to DAFS_HIP_EOVERFLOW when the pool is exhausted */
  dafs_pairhmm3_model model;
} dafs_pairhmm3_args;

int dafs_hipk_pairhmm_plan(uint32_t ntasks, uint32_t max_len1, uint32_t max_len2, dafs_pairhmm_plan* plan);
int dafs_hipk_pairhmm3_launch(const dafs_pairhmm3_args* args, const dafs_pairhmm_plan* plan, void* hip_stream);
/* Host-side model tables: log of the ProbCons defaults, computed with logf like the reference
 * constructor (ProbabilisticModel.h:55-88). */
void dafs_hip_pairhmm3_default_model(dafs_pairhmm3_model* m);
/* residue byte -> class code (wrapper.cpp:157-170: case-insensitive "ACGUTN", else 'other') */
uint8_t dafs_hip_residue_code(char c);

/* ------------------------------------------------------------------------------------------
 * L0: pair-CRF posterior kernel (CONTRAlign 5-state model).
 * Replaces, per pair: CONTRALIGN::CONTRAlign<float>::ComputePosterior
 * (src/contralign/wrapper.cpp:81-97) + CONTRAlign::calculate's dense->sparse step
 * (src/align.cpp:87-106) + transpose_mp + calculate_similarity_score.  Same argument layout and
 * outputs as the ProbCons kernel; the scratch holds five planes (dafs_hipk_pairhmm5_plan).
 * ---------------------------------------------------------------------------------------- */
typedef struct {
  float match[5][5]; /* symbol 0..3 = A C G U, 4 = anything else (scores 0), Defaults.ipp:393-402 */
  float insert[5];   /* Defaults.ipp:403-406 */
  float single[5];   /* per state MATCH, INS_X, INS_Y, INS2_X, INS2_Y, Defaults.ipp:407-409 */
  float pair[5][5];  /* transition [from][to], Defaults.ipp:410-416 */
} dafs_pairhmm5_model;

typedef struct {
  const uint8_t* codes;
  const dafs_pair_task* tasks;
  uint32_t ntasks;
  float th;
  float* scratch;
  uint32_t* queue;
  const uint64_t* rp_off;
  uint32_t* rowptr_pool;
  uint32_t* ent_col;
  float* ent_val;
  unsigned long long* pool_top;
  uint64_t pool_cap;
  uint64_t* pair_off;
  uint32_t* pair_nnz;
  float* sim;
  int* status;
  dafs_pairhmm5_model model;
} dafs_pairhmm5_args;

int dafs_hipk_pairhmm5_plan(uint32_t ntasks, uint32_t max_len1, uint32_t max_len2, dafs_pairhmm_plan* plan);
int dafs_hipk_pairhmm5_launch(const dafs_pairhmm5_args* args, const dafs_pairhmm_plan* plan, void* hip_stream);
void dafs_hip_pairhmm5_default_model(dafs_pairhmm5_model* m);

/* ------------------------------------------------------------------------------------------
 * L1: batch alignment-posterior plugin.
 * Replaces Align::Model::calculate(const vector<Fasta>&, vector<vector<MP>>&)
 * (src/align.cpp:35-52) for -a ProbCons / -a CONTRAlign, plus the transposes (src/dafs.cpp:1797-1799) and
 * sim_ (src/dafs.cpp:1813-1819), for the pair-index shard [pair_begin, pair_end) of the
 * row-major (i<j) pair enumeration (pair_end = 0 means all pairs).
 * ---------------------------------------------------------------------------------------- */
typedef struct dafs_hip_ctx dafs_hip_ctx;
int dafs_hip_create(int device, dafs_hip_ctx** ctx);
void dafs_hip_destroy(dafs_hip_ctx* ctx);

int dafs_hip_set_sequences(dafs_hip_ctx* ctx, uint32_t nseq, const char* const* seqs, const uint32_t* lens);
int dafs_hip_align_posteriors(dafs_hip_ctx* ctx, int model, float th, uint64_t pair_begin, uint64_t pair_end);
/* sizes of the result held in the context */
int dafs_hip_align_result_size(dafs_hip_ctx* ctx, uint64_t* npairs, uint64_t* total_nnz, uint64_t* total_rowptr);
/* Copy out, pairs in shard order p = pair_begin..: pair_x/pair_y[npairs]; sim[npairs];
 * nnz[npairs]; rowptr: for each pair len_x+1 entries (relative) then len_y+1 (transposed);
 * col/val: for each pair nnz entries of mp[x][y] then nnz of mp[y][x]. Any pointer may be NULL. */
int dafs_hip_align_fetch(dafs_hip_ctx* ctx, uint32_t* pair_x, uint32_t* pair_y, float* sim, uint32_t* nnz,
                         uint32_t* rowptr, uint32_t* col, float* val);

/* Generic access to the matching-probability stores: relaxed = 0 as computed by the alignment
 * model, 1 after dafs_hip_consistency (layout as dafs_hip_align_fetch). */
int dafs_hip_mp_result_size(dafs_hip_ctx* ctx, int relaxed, uint64_t* npairs, uint64_t* total_nnz, uint64_t* total_rowptr);
int dafs_hip_mp_fetch(dafs_hip_ctx* ctx, int relaxed, uint32_t* pair_x, uint32_t* pair_y, uint32_t* nnz,
                      uint32_t* rowptr, uint32_t* col, float* val);
/* Supplied matching probabilities instead of a model: AUXAlign::calculate (src/align.cpp:204-246, --align-aux), and
 * the hand-over after an all-gather of shards.  For every pair x < y in row-major order: nnz[p] entries, len[x]+1 row
 * pointers relative to the pair's first entry (concatenated), then all (col ascending within a row, val) entries
 * concatenated.  Lays out the transposes (transpose_mp, src/dafs.cpp:155-167) and computes the similarity scores
 * (calculate_similarity_score, :713-764) on the device. */
int dafs_hip_set_mp(dafs_hip_ctx* ctx, const uint32_t* nnz, const uint32_t* rowptr, const uint32_t* col, const float* val);
/* A whole store from arrays in the layout of dafs_hip_mp_fetch / dafs_hip_align_fetch (both directions of every pair; no
 * recomputation, uploads only): how the ranks of a multi-GPU run take their gathered shards back in.  relaxed = 0: the
 * models' posteriors, with sim[npairs] = the similarity score of every pair (src/dafs.cpp:763) as the shard kernels
 * computed them; relaxed = 1: the consistency transform's result, on top of an un-relaxed store (sim may be NULL). */
int dafs_hip_mp_install(dafs_hip_ctx* ctx, int relaxed, const uint32_t* nnz, const uint32_t* rowptr, const uint32_t* col,
                        const float* val, const float* sim);
/* sim_ (src/dafs.cpp:1813-1819): N*N floats, unit diagonal; needs a full-pair-set align_posteriors, dafs_hip_set_mp or dafs_hip_mp_install. */
int dafs_hip_get_sim(dafs_hip_ctx* ctx, float* sim);

/* ------------------------------------------------------------------------------------------
 * L1: base-pairing probabilities.
 * dafs_hip_set_bp replaces AUXFold::calculate (src/fold.cpp:261-278, --fold-aux): the caller
 * supplies BP rows.  rowptr: per sequence len+1 entries relative to that sequence's first entry,
 * concatenated in input order; col/val: entries (j > i, p) of all sequences concatenated.
 * ---------------------------------------------------------------------------------------- */
int dafs_hip_set_bp(dafs_hip_ctx* ctx, const uint32_t* rowptr, const uint32_t* col, const float* val);
int dafs_hip_bp_result_size(dafs_hip_ctx* ctx, int relaxed, uint64_t* total_nnz, uint64_t* total_rowptr);
int dafs_hip_bp_fetch(dafs_hip_ctx* ctx, int relaxed, uint32_t* rowptr, uint32_t* col, float* val);

/* Batch hook replacing Fold::Model::calculate(const vector<Fasta>&, vector<BP>&)
 * (src/fold.cpp:60-68) for -s CONTRAfold (CONTRAfold::calculate, src/fold.cpp:174-189): inside /
 * outside / posterior of every sequence on the device, rows with p > th kept (reference CUTOFF
 * 0.01, src/dafs.cpp:1704).  Fills the same store dafs_hip_set_bp fills. */
int dafs_hip_fold_posteriors(dafs_hip_ctx* ctx, int model, float th);
/* The same in two halves: _begin enqueues the folding kernels on a stream of their own and returns, _end waits and
 * fills the store.  Between them the calls that do not need base-pairing probabilities may run (the all-pairs
 * alignment posteriors, dafs_hip_consistency_match): the folding occupies one workgroup per sequence, which leaves
 * most of the device idle at N < #CUs. */
int dafs_hip_fold_posteriors_begin(dafs_hip_ctx* ctx, int model, float th);
int dafs_hip_fold_posteriors_end(dafs_hip_ctx* ctx);
/* Single-sequence call replacing CONTRAfold<float>::ComputePosterior (src/contrafold/wrapper.cpp:181-200)
 * and, with a constraint string of len chars from "?.()" ('?' free, '.' unpaired, brackets forced),
 * Fold::Model::calculate(seq, str, bp) (src/fold.cpp:191-207).  post: (len+1)(len+2)/2 floats,
 * upper-triangular rows i = 0..len, columns j = i..len; logz (optional): log partition function. */
int dafs_hip_fold_posterior_dense(dafs_hip_ctx* ctx, const char* seq, uint32_t len, const char* constraint, float* post,
                                  float* logz);

/* Host-side helper (no device work): DAFS::build_tree (src/dafs.cpp:446-492) on the N*N similarity matrix.
 * score/left/right: 2N-1 entries; leaves have left = right = -1, node N+k is the k-th join of slots left, right. */
int dafs_host_build_tree(uint32_t n, const float* sim, float* score, int32_t* left, int32_t* right);

/* ------------------------------------------------------------------------------------------
 * L1: probabilistic consistency transforms.
 * Replaces DAFS::relax_basepairing_probability (src/dafs.cpp:326-375) and
 * DAFS::relax_matching_probability (src/dafs.cpp:258-324) as called from DAFS::run
 * (src/dafs.cpp:1822-1827): both read the un-relaxed stores; weight 0 skips a transform.
 * ---------------------------------------------------------------------------------------- */
int dafs_hip_consistency(dafs_hip_ctx* ctx, float w_pct_a, float w_pct_s);
/* the two transforms separately (each reads un-relaxed stores only) */
int dafs_hip_consistency_match(dafs_hip_ctx* ctx, float w_pct_a);
int dafs_hip_consistency_bp(dafs_hip_ctx* ctx, float w_pct_s);
/* DAFS::relax_fourway_consistency (src/dafs.cpp:377-444; option -f, called at :1808-1809 between the alignment model and
 * the similarity scores): the un-relaxed matching store is replaced by its mix with the stacking evidence of the un-relaxed
 * base-pairing store (dafs_hip_fold_posteriors / dafs_hip_set_bp must have run), and the similarity scores are recomputed
 * from the result.  Call before dafs_hip_get_sim / dafs_hip_consistency*.  w_pct_f = 0 does nothing, like the reference. */
int dafs_hip_fourway_consistency(dafs_hip_ctx* ctx, float w_pct_f);
/* relax_matching_probability for the output pairs [pair_begin, pair_end) of the row-major pair enumeration only (every
 * output pair is independent of the others, src/dafs.cpp:265-315): the shard of one rank of a multi-GPU run.  The
 * other pairs of the relaxed store stay empty until the gathered whole is installed with dafs_hip_mp_install. */
int dafs_hip_consistency_match_range(dafs_hip_ctx* ctx, float w_pct_a, uint64_t pair_begin, uint64_t pair_end);

/* ------------------------------------------------------------------------------------------
 * L1: decoder plugins on dense row-major matrices (host buffers).
 * dafs_hip_nussinov_decode replaces SparseNussinov::decode(w,p,q,ss) (src/nussinov.cpp:207-298);
 * with q == NULL it is the final-decode overload decode(p,ss,str) (:300-392, score p-th, w unused;
 * brackets: dafs_hip_make_brackets).  ss[i] = partner index or DAFS_HIP_NONE.
 * dafs_hip_nw_envelope replaces SparseNeedlemanWunsch::initialize (src/needleman_wunsch.cpp:198-253):
 * env[2*i], env[2*i+1] = first/last column of row i, i = 0..L1.
 * dafs_hip_nw_decode replaces SparseNeedlemanWunsch::decode (:255-422), q may be NULL;
 * al[i] = aligned column or DAFS_HIP_NONE.
 * ---------------------------------------------------------------------------------------- */
int dafs_hip_nussinov_decode(dafs_hip_ctx* ctx, float th, float w, uint32_t L, const float* p, const float* q,
                             uint32_t* ss, float* score);
int dafs_hip_nw_envelope(dafs_hip_ctx* ctx, float th, uint32_t L1, uint32_t L2, const float* p, uint32_t* env);
int dafs_hip_nw_decode(dafs_hip_ctx* ctx, float th, uint32_t L1, uint32_t L2, const float* p, const float* q,
                       const uint32_t* env, uint32_t* al, float* score);
/* The dense decoder classes of the reference (Nussinov, src/nussinov.cpp:32-204: every pair scores, bifurcation over
 * every split; NeedlemanWunsch, src/needleman_wunsch.cpp:28-196: no envelope).  DAFS instantiates the sparse ones
 * (src/dafs.cpp:1692,1759); these complete the plugin set.  q may be NULL (the final-decode overloads). */
int dafs_hip_nussinov_decode_dense(dafs_hip_ctx* ctx, float th, float w, uint32_t L, const float* p, const float* q,
                                   uint32_t* ss, float* score);
int dafs_hip_nw_decode_dense(dafs_hip_ctx* ctx, float th, uint32_t L1, uint32_t L2, const float* p, const float* q,
                             uint32_t* al, float* score);
/* make_brackets (src/nussinov.cpp:401-413): str needs L+1 bytes. Host-only helper. */
void dafs_hip_make_brackets(uint32_t L, const uint32_t* ss, char* str);

/* ------------------------------------------------------------------------------------------
 * L1: fused guide-tree node solver.
 * One call solves a batch of independent nodes: for each, average the base-pairing and matching
 * probabilities of the two child alignments (src/dafs.cpp:513-607, no alifold term), enumerate the
 * consensus base pairs, and run the whole dual-decomposition loop on the device
 * (DAFS::solve_by_dd, src/dafs.cpp:1006-1295).  Uses the context's current stores (after
 * dafs_hip_consistency when that was called).
 * ---------------------------------------------------------------------------------------- */
typedef struct {
  uint32_t n1, n2;        /* rows (sequences) of the two child alignments                     */
  uint32_t len1, len2;    /* their column counts                                              */
  const uint32_t* seq1;   /* [n1] sequence index of each row                                  */
  const uint32_t* seq2;   /* [n2]                                                             */
  const uint8_t* mask1;   /* [n1*len1] 1 = residue, 0 = gap (the reference's vector<bool>)    */
  const uint8_t* mask2;   /* [n2*len2]                                                        */
  const float* p_x;       /* optional [len1*len1]: base-pairing matrix of alignment 1 to use in   */
  const float* p_y;       /* optional [len2*len2]  place of the averaged one (--bp-update: the matrices
                             re-estimated by dafs_hip_update_basepairing, dafs.cpp:919-934); NULL = average */
} dafs_node_input;

typedef struct {
  uint32_t* x;            /* [len1] common structure of alignment 1 (may be NULL)             */
  uint32_t* y;            /* [len2] (may be NULL)                                             */
  uint32_t* z;            /* [len1] column of alignment 2 aligned to each column of 1         */
  float score;            /* value solve_by_dd returns (s_prev)                               */
  uint32_t ncbp, iterations, violated; /* dafs.cpp:1292 log line                             */
} dafs_node_output;

typedef struct {
  float w;                /* -w    default 4.0 */
  float eta0;             /* --eta default 0.5 */
  float th_a;             /* -u    default 0.01 */
  float th_s;             /* -t    default 0.2 */
  uint32_t t_max;         /* -m    default 600 */
  int force_iters;        /* bench only: ignore the violated==0 exit (never for parity runs)  */
  int skip_uncoupled_folds; /* default 0 = solve_by_dd as it is.  1: a node with no consensus base pair (ncbp == 0:
                             nothing couples its three subproblems, one pass, no multiplier ever moves) runs the
                             alignment DP only; x and y come back empty and score holds the alignment part.  For
                             callers that consume z alone, as DAFS::align_alignments does (dafs.cpp:896-912).  */
} dafs_dd_params;
void dafs_hip_dd_default_params(dafs_dd_params* p);
int dafs_hip_solve_nodes(dafs_hip_ctx* ctx, uint32_t nnodes, const dafs_node_input* in, const dafs_dd_params* prm,
                         dafs_node_output* out);

/* The same solver with the nodes resident on the device, so that the progressive phase needs no level
 * barrier: open the nodes whose children are ready, advance all open nodes by at most max_iterations
 * subgradient iterations per call (one launch), collect the finished ones, open their parents, repeat.
 * Replaces the recursion of DAFS::align (reference src/dafs.cpp:983-1004) as the driver of
 * DAFS::align_alignments / solve_by_dd.  Results are those of dafs_hip_solve_nodes, bit for bit.        */
int dafs_hip_nodes_open(dafs_hip_ctx* ctx, uint32_t nnodes, const dafs_node_input* in, const dafs_dd_params* prm, uint32_t* handles);
int dafs_hip_nodes_advance(dafs_hip_ctx* ctx, uint32_t n, const uint32_t* handles, const dafs_dd_params* prm, uint32_t max_iterations,
                           uint8_t* finished);
/* One round in a single call: the n_old open nodes advance while the n_new nodes whose children have just finished are
 * opened and started beside them on a second stream (their handles come back in new_handles).  Every node runs at most
 * max_iterations iterations and, when budget_us > 0, stops at the first iteration end past budget_us microseconds after
 * the round began, late starters included.  Results are those of the calls above, bit for bit.                          */
int dafs_hip_nodes_round(dafs_hip_ctx* ctx, uint32_t n_new, const dafs_node_input* in, uint32_t* new_handles, uint32_t n_old,
                         const uint32_t* old_handles, const dafs_dd_params* prm, uint32_t max_iterations, uint32_t budget_us,
                         uint8_t* finished_old, uint8_t* finished_new);
int dafs_hip_nodes_result(dafs_hip_ctx* ctx, uint32_t handle, dafs_node_output* out);
int dafs_hip_nodes_close(dafs_hip_ctx* ctx);
/* Device memory of the resident nodes (diagnostics): bytes reserved from the device, bytes held by open nodes now, and
 * the largest value the latter has had.  A node's memory is returned when dafs_hip_nodes_result has copied it out. */
int dafs_hip_nodes_memory(dafs_hip_ctx* ctx, uint64_t* reserved, uint64_t* in_use, uint64_t* peak);
/* Split-mode nodes whose folding workgroups did not show up in time and that went on in the one-workgroup form since the
 * last dafs_hip_nodes_close (diagnostics: results are unaffected, the run is slower; 0 on an undisturbed device). */
int dafs_hip_nodes_demotions(dafs_hip_ctx* ctx, uint32_t* n);
/* Final common structure of an alignment (src/dafs.cpp:1857-1871 without the RNAalifold term):
 * averaged base-pairing matrix -> SparseNussinov::decode(p,ss,str) with threshold th. */
int dafs_hip_consensus_structure(dafs_hip_ctx* ctx, uint32_t n, uint32_t len, const uint32_t* seq, const uint8_t* mask,
                                 float th, uint32_t* ss, float* score, float* p_out);

/* DAFS::update_basepairing_probability (src/dafs.cpp:609-712; options --bp-update and --bp-update1), without the RNAalifold
 * term: the sequences of the alignment (rows seq / mask as in dafs_node_input) are folded again with CONTRAfold under the
 * constraint the common structure ss (ss[i] = j for the left partner, DAFS_HIP_NONE otherwise, as the decoders return it)
 * puts on each of them, and the constrained posteriors are averaged; p_out receives the len x len matrix. */
int dafs_hip_update_basepairing(dafs_hip_ctx* ctx, uint32_t n, uint32_t len, const uint32_t* seq, const uint8_t* mask,
                                const uint32_t* ss, float* p_out);

/* ---- device-resident exchange of the sparse stores (multi-GPU runs) ---------------------------------------------------
 * One process per GPU shards phase 1 of DAFS::run (src/dafs.cpp:1787-1827): the folds (src/fold.cpp:66-67), the pair jobs
 * (src/align.cpp:46-50) and the output pairs of relax_matching_probability (src/dafs.cpp:265-315) are independent.  The
 * shards travel by all-gather over RCCL; these entry points move a store to and from DEVICE buffers of the context's own
 * device in the layouts of dafs_hip_mp_fetch / dafs_hip_bp_fetch, so nothing goes through host memory.  Every pointer
 * argument is device memory except where noted.
 *   dafs_hip_mp_export_dev   pairs [first, first + count) of the store's own pair order (an un-relaxed shard computed by
 *                            dafs_hip_align_posteriors(pair_begin, pair_end) numbers its pairs from 0; the relaxed store
 *                            keeps the global row-major index): nnz[count], the pairs' relative row pointers, their entries
 *                            (col / val, capacity cap_entries), and for relaxed = 0 the similarity scores sim[count].
 *                            n_rowptr / n_entries (host) receive what was written.
 *   dafs_hip_mp_install_dev  the whole store from arrays of all N(N-1)/2 pairs in row-major order (dafs_hip_mp_install's
 *                            arguments, on the device); n_entries = entries in col / val = 2 * sum of nnz.
 *   dafs_hip_bp_export_dev   the un-relaxed base-pairing store, sequences in input order (dafs_hip_bp_fetch's layout).
 *   dafs_hip_set_bp_dev      the un-relaxed base-pairing store from nblocks = N blocks in any order: block k holds the rows
 *                            of sequence seq_of_block[k] (host array); rowptr / col / val are the blocks' arrays concatenated
 *                            in block order -- what the ranks' exports look like after the gather. */
int dafs_hip_mp_export_dev(dafs_hip_ctx* ctx, int relaxed, uint64_t first, uint64_t count, uint32_t* nnz, uint32_t* rowptr, uint32_t* col, float* val,
                           float* sim, uint64_t cap_entries, uint64_t* n_rowptr, uint64_t* n_entries);
int dafs_hip_mp_install_dev(dafs_hip_ctx* ctx, int relaxed, const uint32_t* nnz, const uint32_t* rowptr, const uint32_t* col, const float* val,
                            const float* sim, uint64_t n_entries);
int dafs_hip_bp_export_dev(dafs_hip_ctx* ctx, uint32_t* rowptr, uint32_t* col, float* val, uint64_t cap_entries, uint64_t* n_rowptr, uint64_t* n_entries);
int dafs_hip_set_bp_dev(dafs_hip_ctx* ctx, uint32_t nblocks, const uint32_t* seq_of_block, const uint32_t* rowptr, const uint32_t* col, const float* val,
                        uint64_t n_entries);

/* ---- phase 1 of DAFS::run on one rank of a multi-GPU run (src/dafs.cpp:1787-1827) ----
 * One process per GPU; every rank has called dafs_hip_set_sequences with all N sequences.  Rank r folds the sequences
 * x = r (mod world) (independent per sequence, src/fold.cpp:66-67), computes the pair posteriors and similarity scores of
 * the r-th contiguous range of the row-major pair enumeration (dafs_hip_pair_range; independent per pair,
 * src/align.cpp:46-50) and relax_matching_probability's output pairs of the same range (src/dafs.cpp:265-315); after each of
 * the three pieces the shards are packed on the device, all-gathered by `allgather` and installed, so that on return the
 * context is in the state a single-GPU phase 1 (fold_posteriors, align_posteriors, consistency) leaves it in, bit for
 * bit, on every rank.  The base-pairing transform is replicated.  Not for the aux-file inputs or the four-way transform.
 *
 * dafs_allgather_fn: gather `bytes` bytes from every rank (send, device memory) into recv (device memory, world * bytes,
 * rank order) -- ncclAllGather's contract.  The collective may be enqueued on hip_stream (a hipStream_t of the context's
 * device); the library waits for that stream before it reads recv.  Non-zero return = failure (DAFS_HIP_ECOMM). */
typedef int (*dafs_allgather_fn)(void* user, const void* send, void* recv, size_t bytes, void* hip_stream);
void dafs_hip_pair_range(uint64_t npairs, uint32_t world, uint32_t rank, uint64_t* begin, uint64_t* end);
int dafs_hip_phase1_sharded(dafs_hip_ctx* ctx, uint32_t rank, uint32_t world, int align_model, float th_a, float w_pct_a, float w_pct_s,
                            int fold_model, float fold_th, dafs_allgather_fn allgather, void* user);

/* ---- measurement aid: device time per kernel (bench.py's "stages") ----
 * dafs_hip_stage_timing(ctx, 1) makes every kernel launch of the library record a pair of HIP events on its stream
 * (one context per process at a time); dafs_hip_stage_report waits for the device, adds the elapsed times up per kernel and
 * starts the next interval; dafs_hip_stage_timing(ctx, 0) switches it off.  `ms` of kernels that ran beside others on
 * different streams overlap: they are per-kernel durations, not a partition of the wall-clock. */
typedef struct {
  const char* kernel;   /* kernel name as in the rocprofv3 kernel trace (template arguments left out) */
  double ms;            /* sum over the launches of the interval */
  double longest_ms;    /* the longest single launch */
  uint32_t launches;
} dafs_stage_time;
int dafs_hip_stage_timing(dafs_hip_ctx* ctx, int enable);
int dafs_hip_stage_report(dafs_hip_ctx* ctx, dafs_stage_time* out, uint32_t cap, uint32_t* n);

#ifdef __cplusplus
}
#endif
#endif
