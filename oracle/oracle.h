/* oracle/oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C, single thread) of the reference algorithm for the DAFS hot path.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load liboracle.so;
 * the product (dafs_amd/, include/) never links, imports or calls anything declared here.
 *
 * Every function cites the reference file:line it follows (paths relative to /root/reference).
 * Pinning status is stated per group in the .c files and in DESIGN.md.
 */
#ifndef DAFS_ORACLE_H
#define DAFS_ORACLE_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_NONE 0xFFFFFFFFu /* "-1u" of the reference (e.g. src/nussinov.cpp:267) */

/* Sparse row lists (reference MP / BP = vector<vector<pair<uint,float>>>, src/typedefs.h:37-39)
 * flattened as CSR.  Owned by the oracle; free with orc_csr_free. */
typedef struct {
  uint32_t nrow;
  uint32_t* rowptr; /* nrow+1 */
  uint32_t* col;
  float* val;
} orc_csr;
void orc_csr_free(orc_csr* m);

/* ---- ProbCons pair-HMM (src/probconsRNA) ---- */
/* dense posterior, (L1+1)*(L2+1) floats, entries < th zeroed (wrapper.cpp:101-131) */
int orc_probcons_posterior(const char* s1, uint32_t L1, const char* s2, uint32_t L2, float th, float* out);

/* ---- CONTRAlign 5-state pair CRF (src/contralign) ---- */
int orc_contralign_posterior(const char* s1, uint32_t L1, const char* s2, uint32_t L2, float th, float* out);

/* ---- Align::Model adapter: dense -> MP rows with p > th (src/align.cpp:60-106) ----
 * model 0 = ProbCons, 1 = CONTRAlign.  Caller buffers: rowptr[L1+1], col/val[L1*L2]. */
int orc_align_calculate(int model, const char* s1, uint32_t L1, const char* s2, uint32_t L2, float th,
                        uint32_t* rowptr, uint32_t* col, float* val);

/* ---- CONTRAfold (src/contrafold) ---- */
/* upper-triangular posterior, (L+1)(L+2)/2 floats; constraint NULL or L chars of "?.()" */
int orc_contrafold_posterior(const char* seq, uint32_t L, const char* constraint, float* out);
float orc_contrafold_logz(const char* seq, uint32_t L);
/* Fold::Model adapter: triangular -> BP rows with p > th (src/fold.cpp:174-207) */
int orc_fold_calculate(const char* seq, uint32_t L, const char* constraint, float th,
                       uint32_t* rowptr, uint32_t* col, float* val);

/* ---- decoders ---- */
/* SparseNussinov::decode(w,p,q,ss) src/nussinov.cpp:207-298; q==NULL selects the 3-arg
 * final-decode twin (:300-392) whose score is p-th (w ignored). p,q row-major L*L. */
float orc_nussinov_decode(float th, float w, uint32_t L, const float* p, const float* q, uint32_t* ss);
void orc_make_brackets(uint32_t L, const uint32_t* ss, char* str /* L+1 */);
/* SparseNeedlemanWunsch::initialize src/needleman_wunsch.cpp:198-253; env[2*(L1+1)] = first,second */
/* dense classes Nussinov / NeedlemanWunsch (src/nussinov.cpp:32-204, src/needleman_wunsch.cpp:28-196); q may be NULL */
float orc_nussinov_dense_decode(float th, float w, uint32_t L, const float* p, const float* q, uint32_t* ss);
float orc_nw_dense_decode(float th, uint32_t L1, uint32_t L2, const float* p, const float* q, uint32_t* al);
void orc_nw_envelope(float th, uint32_t L1, uint32_t L2, const float* p, uint32_t* env);
/* SparseNeedlemanWunsch::decode :255-422; q may be NULL */
float orc_nw_decode(float th, uint32_t L1, uint32_t L2, const float* p, const float* q,
                    const uint32_t* env, uint32_t* al);

/* ---- pieces of src/dafs.cpp ---- */
/* calculate_similarity_score dafs.cpp:713-764 */
float orc_similarity_score(const uint32_t* rowptr, const uint32_t* col, const float* val, uint32_t L1, uint32_t L2);
/* transpose_mp dafs.cpp:155-167 */
void orc_transpose(const orc_csr* in, uint32_t ncol, orc_csr* out);

/* Whole pipeline (DAFS::run dafs.cpp:1781-1889) on in-memory sequences. */
typedef struct {
  int align_model;   /* 0 ProbCons, 1 CONTRAlign (-a); 2 = mp supplied via orc_pipeline_set_mp (--align-aux) */
  int fold_model;    /* 0 CONTRAfold, 1 = bp supplied via orc_pipeline_set_bp (--fold-aux) */
  float w;           /* -w   default 4.0  */
  float eta0;        /* --eta default 0.5 */
  uint32_t t_max;    /* -m   default 600  */
  float w_pct_a;     /* -p   default 0.25 */
  float w_pct_s;     /* -q   default 0.25 */
  float th_a;        /* -u   default 0.01 */
  float th_s;        /* -t   default 0.2  */
  float th_s1;       /* -T   default = th_s */
  int force_iters;   /* bench-only: ignore the violated==0 exit (never for parity) */
  float w_pct_f;     /* -f   default 0.0: weight of the four-way consistency transform (dafs.cpp:377-444, :1808) */
  int bp_update;     /* --bp-update  (dafs.cpp:1766): base-pairing matrices of the root node re-estimated under the decoded structure */
  int bp_update1;    /* --bp-update1 (dafs.cpp:1767): the same for the final common structure */
} orc_params;
void orc_params_default(orc_params* p);

typedef struct orc_pipeline orc_pipeline;
orc_pipeline* orc_pipeline_new(const orc_params* prm, uint32_t N, const char* const* names, const char* const* seqs);
void orc_pipeline_free(orc_pipeline* pl);
/* --fold-aux equivalent: inject BP for sequence x (src/fold.cpp:230-278) */
void orc_pipeline_set_bp(orc_pipeline* pl, uint32_t x, const uint32_t* rowptr, const uint32_t* col, const float* val);
/* --align-aux equivalent: inject the rows of mp[x][y], x < y (src/align.cpp:190-247); needs prm.align_model == 2 */
void orc_pipeline_set_mp(orc_pipeline* pl, uint32_t x, uint32_t y, const uint32_t* rowptr, const uint32_t* col, const float* val);
/* phase 1: bp_, mp_ (+transposes), sim_, PCTs, tree  (dafs.cpp:1787-1830) */
int orc_pipeline_phase1(orc_pipeline* pl);
/* phase 2: progressive alignment + final SS (dafs.cpp:1835-1876) */
int orc_pipeline_phase2(orc_pipeline* pl);
/* accessors (pointers stay owned by the pipeline) */
const orc_csr* orc_pipeline_mp(const orc_pipeline* pl, uint32_t x, uint32_t y);
const orc_csr* orc_pipeline_bp(const orc_pipeline* pl, uint32_t x);
const float* orc_pipeline_sim(const orc_pipeline* pl);          /* N*N */
/* tree: 2N-1 nodes: score, left, right (ORC_NONE for leaves) dafs.cpp:446-492 */
void orc_pipeline_tree(const orc_pipeline* pl, float* score, uint32_t* left, uint32_t* right);
/* text exactly as DAFS::run prints it (tree line, >SS_cons, bracket string, rows) */
const char* orc_pipeline_output(orc_pipeline* pl);
/* per internal node (in solve order): iterations run and final violated count (dafs.cpp:1292) */
uint32_t orc_pipeline_dd_log(const orc_pipeline* pl, uint32_t* iters, uint32_t* violated, uint32_t cap);
double orc_pipeline_seconds(const orc_pipeline* pl, int phase); /* 0 fold,1 pair,2 pct+tree,3 progressive */

/* FASTA reader following Fasta::load src/fa.cpp:37-87. Returns count or -1.
 * names/seqs are malloc'ed arrays of malloc'ed strings (free with orc_fasta_free). */
int orc_fasta_load(const char* file, char*** names, char*** seqs);
void orc_fasta_free(int n, char** names, char** seqs);

#ifdef __cplusplus
}
#endif
#endif
