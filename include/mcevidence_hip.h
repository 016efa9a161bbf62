/* mcevidence_hip.h -- C ABI of libmcevidence_hip.so (MI355X / gfx950).
 *
 * Drop-in boundary for the k-nearest-neighbour evidence hot path of
 * yabebalFantaye/MCEvidence.  The reference has no FFI of its own: the seam is
 * the Python call into scikit-learn plus ~25 lines of NumPy
 * (reference MCEvidence.py:1093-1131).  Each entry point below names the
 * reference lines it replaces.  The Python binding a maintainer would add is
 * shown in INTEGRATION.md (ctypes; mcevidence_amd/_capi.py is that binding).
 *
 * Conventions
 *   - plain C types only; every matrix is C-contiguous (row-major) fp64.
 *   - *_f64      : pointers are HOST pointers to caller-owned buffers; the
 *                  library owns every device allocation it makes and frees it
 *                  before returning; nothing is retained after return.
 *   - *_f64_dev  : pointers are DEVICE pointers on the current HIP device
 *                  (e.g. torch tensors' data_ptr()); work is enqueued on
 *                  `stream` (a hipStream_t passed as void*, NULL = default
 *                  stream) and NOT synchronised; `ws` is caller-provided
 *                  scratch of at least mce_*_workspace_bytes().
 *   - return 0 on success, a negative MCE_ERR_* otherwise; mce_last_error()
 *     returns a thread-local message for the last failure on this thread.
 *   - one caller thread per device; calls are re-entrant across devices.  Several threads may share a device; the
 *     search / prune / symmetric modes then travel WITH the call (mce_options), not through the process-wide setters.
 *
 * Neighbour semantics (all entry points): Euclidean distance, the K smallest
 * per query in ascending order, ties broken by smaller reference index --
 * what `NearestNeighbors(...).kneighbors()` returns at MCEvidence.py:1104.
 */
#ifndef MCEVIDENCE_HIP_H
#define MCEVIDENCE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MCE_ABI_VERSION 3     /* 2: per-call options (mce_options, *_opt), mce_last_search_stats; mce_options.verify and mce_verify_* were
                               * added compatibly (the field lies in what was reserved[0] = 0).
                               * 3 (round 6): every signature of 2 unchanged; NEW entry points (mce_last_verify_rows,
                               * mce_evidence_feed_part_dev_f64, mce_evidence_feed_whiten_dev_f64) and one changed DEFAULT:
                               * mce_options.verify = -1 now means "256 rows behind the fp16 filter" (0 still means off) */

#define MCE_OK 0
#define MCE_ERR_INVALID (-1)   /* bad argument (NULL, d<1, K<1, ...)        -> ValueError  */
#define MCE_ERR_K_RANGE (-2)   /* K larger than usable reference rows / MCE_MAX_K -> ValueError */
#define MCE_ERR_HIP (-3)       /* HIP runtime failure                        -> RuntimeError */
#define MCE_ERR_NO_DEVICE (-4) /* no gfx950 device visible                   -> RuntimeError */
#define MCE_ERR_WORKSPACE (-5) /* caller workspace too small                 -> ValueError  */
#define MCE_ERR_DIM_RANGE (-6) /* d larger than MCE_MAX_DIM                  -> ValueError  */
#define MCE_ERR_VERIFY (-7)    /* the re-check of sampled rows disagrees with the search (mce_options.verify) -> RuntimeError */

#define MCE_MAX_K 32    /* neighbours per query handled by the MFMA kernels (fp16 filter: 17..32 in two sweeps) */
#define MCE_MAX_DIM 63  /* dimensions handled by the one-to-four k-step fp16 filter (64..127: the deep fp16 filter, K <= 32 -- search mode 1: the
                           fp64 sweep; 128..1024: the fp64 sweep with the k dimension in blocks; feeders: d <= 127) */
#define MCE_GENERIC_MAX_K 1024   /* beyond MCE_MAX_K a plain exact kernel takes over, up to this many neighbours; rows up to */
#define MCE_GENERIC_MAX_DIM 1024 /* this long; larger -> MCE_ERR_K_RANGE / MCE_ERR_DIM_RANGE                                 */

/* self_mode: how the query set relates to the reference set.
 *   MCE_SELF_NONE    Y is a different set (cross evidence, MCEvidence.py:1093-1096, k0=0)
 *   MCE_SELF_INCLUDE Y row (self_offset+q) IS query q; it is reported with distance
 *                    exactly 0 in column 0 (what the auto path sees, MCEvidence.py:1099-1104)
 *   MCE_SELF_EXCLUDE same relation, but that row is skipped: the K columns are true
 *                    neighbours (equivalent to dropping column 0, `k0=1`, :1099)
 * self_offset is the reference row of query 0; query shards of one chain pass
 * their first global row (multi-GPU query sharding). */
#define MCE_SELF_NONE 0
#define MCE_SELF_INCLUDE 1
#define MCE_SELF_EXCLUDE 2

int mce_abi_version(void);
int mce_device_count(void);
const char *mce_last_error(void);

/* ---- host-pointer entry points (the literal drop-in) ------------------- */

/* Replaces NearestNeighbors(n_neighbors=K,...).fit(Y).kneighbors(X)
 * (MCEvidence.py:1093-1104).  dist[nq*K] ascending per row; idx[nq*K] int64 or NULL. */
int mce_knn_f64(const double *X, int64_t nq, const double *Y, int64_t nr, int32_t d, int32_t K,
                int32_t self_mode, int64_t self_offset, double *dist, int64_t *idx, int32_t device);

/* Replaces the volume loop + np.dot (MCEvidence.py:1107-1117):
 * dotp[k] = sum_j pi^(d/2) dist[j*ld+k]^d / Gamma(1+d/2) / w[j] * exp(fs[j]),  k in [k0,kmax).
 * Entries dotp[0..k0) are set to 0. */
int mce_dotp_f64(const double *dist, int64_t nq, int32_t ld, int32_t k0, int32_t kmax, int32_t d,
                 const double *w, const double *fs, double *dotp, int32_t device);

/* Fused path: search + reduction without returning the distances
 * (MCEvidence.py:1093-1117 in one call).  k0 = 1 -> auto evidence (Y must be the
 * set X was cut from; self excluded by index), k0 = 0 -> cross evidence.
 * dotp[kmax] as above.  dist_out (optional, may be NULL) receives the
 * [nq, kmax-k0] neighbour distances that entered the sum (columns k0..kmax-1 of
 * the reference's DkNN).  With ndev > 1 (devices[i] = HIP ordinal; NULL/0 -> device 0 only) the work is split over the
 * devices -- auto evidence of one set: the library's partition (mce_knn_dotp_part_f64), otherwise equal ranges of the
 * query rows -- by one host thread per device, and the partial sums are added ON THE HOST in device order (bitwise
 * reproducible run to run FOR A GIVEN DEVICE COUNT: the partition, hence the order of the sum, changes with the count;
 * SURVEY.md 8e's "alternatively").  The library itself has no RCCL dependency:
 * the one collective of a multi-PROCESS run -- an all-reduce of kmax doubles -- is the caller's
 * (mcevidence_amd/parallel.py: torch.distributed over RCCL), on the partial sums mce_knn_dotp_part_f64 returns. */
int mce_knn_dotp_f64(const double *X, int64_t nq, const double *Y, int64_t nr, int32_t d, int32_t kmax,
                     int32_t k0, int64_t self_offset, const double *w, const double *fs, double *dotp,
                     double *dist_out, const int32_t *devices, int32_t ndev);

/* Whole evidence() inner block with the feeders on the device too: replaces
 * get_covariance (MCEvidence.py:851-882), diagonalise_chain (:842-849) AND :1093-1117 in one call,
 * with ONE upload of the parameter columns.
 *   S1[n1, ld1] / S2[n2, ld2]: raw (un-whitened) parameter rows, row stride ld >= d, first d
 *       columns used.  S2 = NULL -> auto evidence (k0 = 1); otherwise cross evidence (k0 = 0).
 *   cov_mode 0 ("all"): covariance of the rows of S1 and S2 together; 1 ("single"): S1 whitened with
 *       its own eigen-system, S2 with ITS own (the reference's behaviour, :1080-1086).
 *       (Eigen-systems are canonical: eigenvalues descending, each eigenvector's largest component
 *       positive.  With cov_mode 1 AND S2 the result depends on that convention -- as the reference's
 *       depends on LAPACK's -- so the Python class keeps np.linalg.eig for that one combination.)
 *   Outputs: dotp[kmax]; *jacobian = sqrt(det cov) (of S1's covariance in mode 1);
 *       eigenvalues[d] (may be NULL), descending.  A non-positive eigenvalue -> MCE_ERR_INVALID (the reference
 *       raises ValueError: math domain error). */
int mce_evidence_feed_f64(const double *S1, int64_t n1, int64_t ld1, const double *S2, int64_t n2, int64_t ld2,
                          int32_t d, int32_t cov_mode, int32_t kmax, const double *w, const double *fs,
                          double *dotp, double *jacobian, double *eigenvalues, int32_t device);

/* One rank's share of mce_evidence_feed_f64 (multi-GPU, one process per GPU: SURVEY.md 8e; reference
 * MCEvidence.py:1034-1131): every rank passes the SAME arrays; this rank uploads them once, computes covariance, eigen-system
 * and whitening on its device like the single-rank call (identical on every rank), and searches only its share -- auto
 * evidence: the library's partition (mce_knn_dotp_part_f64); cross evidence: the rows [n1 part / nparts, n1 (part + 1) /
 * nparts) of S1 against all of S2.  dotp_part[kmax] = this rank's partial sums: ONE all-reduce(sum) over the ranks -- the
 * caller's (parallel.py: RCCL) -- completes them; jacobian / eigenvalues as in mce_evidence_feed_f64.
 * *checksum (may be NULL): a 64-bit fingerprint of the uploaded rows, weights and fs computed ON THE DEVICE (one extra
 * pass at HBM speed) -- ranks that were handed different samples are detected by comparing it, without hashing on the host. */
int mce_evidence_feed_part_f64(const double *S1, int64_t n1, int64_t ld1, const double *S2, int64_t n2, int64_t ld2,
                               int32_t d, int32_t cov_mode, int32_t kmax, const double *w, const double *fs,
                               int32_t part, int32_t nparts, double *dotp_part, double *jacobian, double *eigenvalues,
                               uint64_t *checksum, int32_t device);
/* Distributed k-d preparation of the pruned walk (round 6; reference: the `fit` of MCEvidence.py:1100-1101, which every rank of a
 * multi-GPU run repeated in full).  Rank `part` of nparts = 2, 4, 8, ... calls mce_prune_part_prepare_dev on the workspace it will
 * search with: the sorts that settle the tree's top log2(nparts) levels run over all rows, everything below them over the rank's
 * own subtree only.  On return *perm_count int32 values at ws + *perm_offset hold the final k-d order inside the rank's range
 * [*seg_lo, *seg_hi) (positions; the ranks' ranges follow each other in rank order and tile the array) and ZEROS elsewhere: an
 * all_gather of the ranges -- or one all-reduce(SUM) of the whole array -- over the ranks (the host's: torch.distributed over RCCL)
 * gives every rank the whole permutation -- bit for bit the single-GPU one -- and mce_knn_dotp_part_prepared_f64_dev then runs the rank's share of the
 * search like mce_knn_dotp_part_f64_dev, minus the sorts.  *perm_count = 0: nothing to exchange (the shape does not take the
 * pruned walk, nparts is not a power of two, the tree is too shallow) -- call mce_knn_dotp_part_f64_dev as before. */
int mce_prune_part_prepare_dev(const double *dY, int64_t nr, int32_t d, int32_t kmax, int32_t part, int32_t nparts,
                               size_t *perm_offset, int64_t *perm_count, int64_t *seg_lo, int64_t *seg_hi, void *ws,
                               size_t ws_bytes, void *stream);
/* 1: an auto-evidence search of this shape on nparts ranks takes the pruned walk AND its preparation can be distributed (plan only,
 * no device work): what a host asks before it chooses the three-step route */
int32_t mce_prune_part_applies(int64_t nr, int32_t d, int32_t kmax, int32_t nparts);
int mce_knn_dotp_part_prepared_f64_dev(const double *dY, int64_t nr, int32_t d, int32_t kmax, int32_t part, int32_t nparts,
                                       const double *d_w, const double *d_fs, double *d_dotp, void *ws, size_t ws_bytes,
                                       void *stream);

/* The same with the inputs ON THE DEVICE already (round 6; SURVEY.md 5: "one H2D + broadcast over xGMI instead of 8 PCIe
 * copies"): dS1 / dS2 / d_w / d_fs are device pointers on `device` (rows ld1 / ld2 doubles apart), produced on a stream the
 * caller has synchronised -- parallel.py uploads 1/W of the chain per rank and all_gathers it over RCCL.  Everything else
 * (host outputs included) as mce_evidence_feed_part_f64; the inputs are copied, not modified. */
int mce_evidence_feed_part_dev_f64(const double *dS1, int64_t n1, int64_t ld1, const double *dS2, int64_t n2, int64_t ld2,
                                   int32_t d, int32_t cov_mode, int32_t kmax, const double *d_w, const double *d_fs,
                                   int32_t part, int32_t nparts, double *dotp_part, double *jacobian, double *eigenvalues,
                                   uint64_t *checksum, int32_t device);
/* The feeders alone (auto evidence): upload, covariance of the n1 rows, Jacobi eigen-system, whitening -- and the whitened rows,
 * the weights and the likelihood terms left in DEVICE buffers of the caller's ([n1, d], [n1], [n1] doubles on `device`) instead of
 * being searched.  For hosts that run the search in several calls with collectives in between (the all-pairs-once partition:
 * parallel.py).  jacobian / eigenvalues / checksum as mce_evidence_feed_part_f64.  Reference: MCEvidence.py:1034-1060. */
int mce_evidence_feed_whiten_f64(const double *S1, int64_t n1, int64_t ld1, int32_t d, int32_t kmax, const double *w,
                                 const double *fs, double *d_X_out, double *d_w_out, double *d_fs_out, double *jacobian,
                                 double *eigenvalues, uint64_t *checksum, int32_t device);
/* ... with the inputs on the device already (as mce_evidence_feed_part_dev_f64) */
int mce_evidence_feed_whiten_dev_f64(const double *dS1, int64_t n1, int64_t ld1, int32_t d, int32_t kmax, const double *d_w,
                                     const double *d_fs, double *d_X_out, double *d_w_out, double *d_fs_out, double *jacobian,
                                     double *eigenvalues, uint64_t *checksum, int32_t device);

/* Many independent evidence problems in one call: the reference's Planck driver
 * runs MCEvidence(...).evidence() once per (data set, model, chain) -- ~600-2400 chains of 6k-100k
 * rows, D = 6-8 -- farmed over MPI ranks (planck_mcevidence.py:306-348).  One such chain fills a
 * fraction of the device, so here the problems of a batch are pipelined over several HIP streams
 * with two host synchronisations per batch.  Each problem has exactly the semantics of
 * mce_evidence_feed_f64 and yields bit-identical dotp / jacobian / eigenvalues.
 *   in : S1,n1,ld1,S2,n2,ld2,d,cov_mode,kmax,w,fs   (as mce_evidence_feed_f64; S2 = NULL -> auto)
 *   out: dotp[kmax], eigenvalues[d] (may be NULL), jacobian, status (MCE_OK or MCE_ERR_*)
 * Problems are spread over `devices` (NULL/0 -> device 0) balanced by n1*nr.  A failing problem does
 * not stop the others; the return value is the status of the first failing problem (its message in
 * mce_last_error(), prefixed "problem <i>: "), or MCE_OK. */
typedef struct mce_feed_problem {
    const double *S1; int64_t n1, ld1;
    const double *S2; int64_t n2, ld2;
    int32_t d, cov_mode, kmax, status;
    const double *w, *fs;
    double *dotp;
    double *eigenvalues;
    double jacobian;
} mce_feed_problem;

int mce_evidence_feed_batch_f64(mce_feed_problem *problems, int64_t nprob, const int32_t *devices, int32_t ndev);
size_t mce_feed_problem_size(void);   /* sizeof(mce_feed_problem) as built: lets a binding check its struct layout */

/* ---- device-pointer entry points (resident data, caller's stream) ------ */

size_t mce_knn_workspace_bytes(int64_t nq, int64_t nr, int32_t d, int32_t K);

int mce_knn_f64_dev(const double *dX, int64_t nq, const double *dY, int64_t nr, int32_t d, int32_t K,
                    int32_t self_mode, int64_t self_offset, double *d_dist, int64_t *d_idx, void *ws,
                    size_t ws_bytes, void *stream);

size_t mce_dotp_workspace_bytes(int64_t nq, int32_t kmax);

int mce_dotp_f64_dev(const double *d_dist, int64_t nq, int32_t ld, int32_t k0, int32_t kmax, int32_t d,
                     const double *d_w, const double *d_fs, double *d_dotp, void *ws, size_t ws_bytes,
                     void *stream);

/* Fused search + reduction on resident data; d_dotp[kmax] device, d_dist_out optional.
 * workspace: mce_knn_workspace_bytes(nq, nr, d, kmax-k0) + mce_dotp_workspace_bytes(nq, kmax). */
int mce_knn_dotp_f64_dev(const double *dX, int64_t nq, const double *dY, int64_t nr, int32_t d, int32_t kmax,
                         int32_t k0, int64_t self_offset, const double *d_w, const double *d_fs,
                         double *d_dotp, double *d_dist_out, void *ws, size_t ws_bytes, void *stream);

/* Multi-GPU building block for the auto evidence (queries = references, k0 = 1): the partial sums over
 * part `part` of `nparts` of the queries.  The LIBRARY chooses the partition -- contiguous rows for the
 * exhaustive sweep (the query shard of SURVEY.md section 8e; up to four parts of a set large enough for the symmetric sweep:
 * ranges of its sorted blocks, see mce_set_sym_mode), every nparts-th WAVE (64 queries: two k-d cells) of the pruned walk's
 * dispatch order, heaviest first (spatially compact work units, one shared ordering, statistically equal shares; a row-range
 * shard would go through the much less efficient separate-sets path; the heaviest waves of a part are served by several
 * workgroups each) -- the parts are disjoint and cover every row, so adding the
 * nparts results gives mce_knn_dotp_f64_dev's dotp up to summation order.  d_w / d_fs: all nr entries.
 * workspace: mce_knn_workspace_bytes(nr, nr, d, kmax-1) + mce_dotp_workspace_bytes(nr, kmax) -- the whole set's, whatever the
 * part (a row shard that would plan more reference splits than fit takes fewer). */
int mce_knn_dotp_part_f64_dev(const double *dY, int64_t nr, int32_t d, int32_t kmax, int32_t part, int32_t nparts,
                              const double *d_w, const double *d_fs, double *d_dotp, void *ws, size_t ws_bytes,
                              void *stream);
/* the same from host buffers (upload, compute, download) */
int mce_knn_dotp_part_f64(const double *Y, int64_t nr, int32_t d, int32_t kmax, int32_t part, int32_t nparts,
                          const double *w, const double *fs, double *dotp, int32_t device);

/* The ALL-PAIRS-ONCE partition of the same sums (round 5; optional -- parallel.py takes it when MCE_PAIRS_ONCE=1): where one GPU
 * would run the one-pass symmetric sweep (mce_pairs_once_blocks > 0), W ranks can multiply every pair of rows once per NODE
 * instead of once per side: rank r owns the sorted 512-row blocks r, r + W, ... and runs the single-GPU units of those blocks
 * (a block against every block below it, both gates on), and ships the candidates it found for rows it does not own to their owners.  Four calls on one workspace (sized as for
 * mce_knn_dotp_part_f64_dev), the collectives between them are the caller's (no RCCL dependency in this library):
 *   prepare -> *bounds_offset / *bounds_count: where in the workspace (bytes from its start) the rows' bounds lie -- doubles,
 *              +inf for the rows of other ranks: all-reduce them with MIN (every rank bounds the rows of its own blocks only);
 *   sweep   -> d_counts[nparts]: 16-byte candidates for every rank (own entry 0); d_flags[blocks]: blocks of other ranks whose
 *              candidates did not fit here (the owner searches such a block again: all-reduce the flags with MAX);
 *   export  -> d_send: the candidates, densely, ordered by destination rank (sum of d_counts entries of 16 bytes);
 *              exchange them (all_to_all with the counts as split sizes);
 *   finish  <- d_recv / nrecv: what arrived; d_flags: the reduced flags; -> d_dotp[kmax]: the sums over this rank's own rows
 *              (NaN if an entry arrived for a row the rank does not own: ranks that disagree about the partition).
 * Adding the nparts results gives mce_knn_dotp_f64_dev's dotp up to summation order.  Replaces the same reference lines as
 * mce_knn_dotp_f64_dev (MCEvidence.py:1093-1117) for one rank's rows. */
int32_t mce_pairs_once_blocks(int64_t nr, int32_t d, int32_t kmax);
/* the workspace of the three calls below (0: the partition does not exist for this shape): the search's and the reduction's, plus
 * one list set per chain of units when a rank's blocks are dealt to several chains (capi_apo.hpp: PairsOnceShape) */
size_t mce_pairs_once_workspace_bytes(int64_t nr, int32_t d, int32_t kmax, int32_t nparts);
int mce_pairs_once_prepare_dev(const double *dY, int64_t nr, int32_t d, int32_t kmax, int32_t part, int32_t nparts,
                               size_t *bounds_offset, int64_t *bounds_count, void *ws, size_t ws_bytes, void *stream);
int mce_pairs_once_sweep_dev(const double *dY, int64_t nr, int32_t d, int32_t kmax, int32_t part, int32_t nparts,
                             int64_t *d_counts, int32_t *d_flags, void *ws, size_t ws_bytes, void *stream);
int mce_pairs_once_export_dev(int64_t nr, int32_t d, int32_t kmax, int32_t part, int32_t nparts, void *d_send, void *ws,
                              size_t ws_bytes, void *stream);
int mce_pairs_once_finish_dev(const double *dY, int64_t nr, int32_t d, int32_t kmax, int32_t part, int32_t nparts,
                              const double *d_w, const double *d_fs, const void *d_recv, int64_t nrecv,
                              const int32_t *d_flags, double *d_dotp, void *ws, size_t ws_bytes, void *stream);

/* Name of the dominant kernel last launched by this thread and its launch
 * geometry (for bench.py / profiles): "knn_mfma_f64<KS=7,KCAP=12>" etc. */
const char *mce_last_kernel(void);
/* rows that the run-time certificate (mce_options.verify) of the most recent host-pointer search re-checked; 0: none ran */
int32_t mce_last_verify_rows(void);
/* SHA-256 (hex) of the kernel sources this library was built from (every .hpp and .hip file of csrc/ in name order; csrc/Makefile).
 * A committed rocprofv3 profile carries the same digest (profiles/<round>/meta.json): bench.py only quotes counters of a
 * profile that was taken from THESE sources. */
const char *mce_source_hash(void);

/* The host-pointer entry points keep their small device buffers (<= 64 MB each) in a per-thread
 * pool between calls; this frees them. */
void mce_release_device_memory(void);

/* Search algorithm.  0 (default) / 2: fp16-MFMA filter with exact fp64 refinement where the
 * shape allows it (all d <= 63, K <= 32), otherwise the fp64 MFMA sweep; 1: always the
 * fp64 MFMA sweep.  Both return the exact fp64 neighbours and distances.  Process-wide. */
int mce_set_search_mode(int mode);
int mce_get_search_mode(void);

/* Per-call options.  The three mode setters (this one, mce_set_prune_mode, mce_set_sym_mode) set PROCESS-WIDE DEFAULTS;
 * two threads with different needs would race on them.  A call carries its own modes instead:
 *   - the *_opt entry points below take a trailing `const mce_options*` (NULL: the defaults);
 *   - mce_options_push(&o) ... mce_options_pop() bracket any other entry point: the pushed modes apply to the calls THIS
 *     THREAD makes in between (they nest; threads the library starts itself, one per device, inherit them).
 * A mode of -1 means "the process default".  `size` = sizeof(mce_options) as the CALLER was built (room to grow): at least
 * 16 (size + the three modes); a field beyond `size` is never read and counts as -1. */
typedef struct mce_options {
    int32_t size;
    int32_t search_mode;   /* as mce_set_search_mode */
    int32_t prune_mode;    /* as mce_set_prune_mode  */
    int32_t sym_mode;      /* as mce_set_sym_mode    */
    int32_t same_set;      /* workspace queries: 1 = X and Y will be ONE buffer (auto evidence), 0 = they will not (no scratch
                              for the symmetric sweep is reserved: ~1.7 GB at 1 M rows, K = 9), -1 = unknown (reserved) */
    int32_t verify;        /* > 0: after the search, re-check this many query rows (spread over the set) by an independent exact
                              fp64 scan of ALL reference rows (mce_verify_knn_f64_dev below) and fail with MCE_ERR_VERIFY if a row
                              disagrees; honoured by the host-pointer entry points (mce_knn_f64[_opt], mce_knn_dotp_f64[_opt],
                              mce_evidence_feed[_batch]_f64 -- not by a rank's share, *_part_*) at d <= 128, K <= 32 (other shapes
                              run on exact fp64 kernels and are skipped silently).  0: off.  -1 (the default, round 6): a search
                              that went through the fp16 FILTER is re-checked on 256 rows -- every call from 65 536 query rows
                              on, one call in eight of the smaller ones (there the check is launch overhead, not rows: 0.16 of
                              0.68 ms at 7 k x 6) -- and one on the fp64 kernels is not; MCE_VERIFY=n in the environment: n rows
                              on EVERY call; MCE_VERIFY=0: off.  ~1 ms at 1 M x 27 with the distances it needs written out */
    int32_t reserved[2];   /* 0 */
} mce_options;
int mce_options_push(const mce_options* opt);
int mce_options_pop(void);
int mce_knn_f64_opt(const double* X, int64_t nq, const double* Y, int64_t nr, int32_t d, int32_t K, int32_t self_mode,
                    int64_t self_offset, double* dist, int64_t* idx, int32_t device, const mce_options* opt);
int mce_knn_dotp_f64_opt(const double* X, int64_t nq, const double* Y, int64_t nr, int32_t d, int32_t kmax, int32_t k0,
                         int64_t self_offset, const double* w, const double* fs, double* dotp, double* dist_out,
                         const int32_t* devices, int32_t ndev, const mce_options* opt);
int mce_knn_dotp_f64_dev_opt(const double* dX, int64_t nq, const double* dY, int64_t nr, int32_t d, int32_t kmax, int32_t k0,
                             int64_t self_offset, const double* d_w, const double* d_fs, double* d_dotp, double* d_dist_out,
                             void* ws, size_t ws_bytes, void* stream, const mce_options* opt);
size_t mce_knn_workspace_bytes_opt(int64_t nq, int64_t nr, int32_t d, int32_t K, const mce_options* opt);

/* Run-time certificate of a finished search (reference: the result of `nbrs.kneighbors(samples)`, MCEvidence.py:1104, whose
 * exactness everything downstream rests on).  For `nsample` query rows spread evenly over the set (the pattern shifted by
 * `seed`) every reference row's squared distance is recomputed by plain fp64 differences -- none of the search's machinery:
 * no matrix cores, no packed operands, no bounds, no lists -- and counted against the K distances the search reported for
 * that row (`dist`, [nq][ld] as mce_knn_f64 / dist_out write them): nobody outside the list may be closer than its k-th
 * entry, and at least k rows must lie within it (relative tolerance 1e-9 on the squared distance).  d <= 128, K <= 32.
 * _dev: device pointers, the caller's stream, never synchronises; `d_result` (device, 2 ints) receives {rows checked, rows
 * that failed}; workspace from mce_verify_workspace_bytes.  The host-pointer form uploads, checks, and returns
 * MCE_ERR_VERIFY (message: how many rows failed) or MCE_OK; `failed` (optional) receives the count. */
size_t mce_verify_workspace_bytes(int32_t nsample, int32_t K);
int mce_verify_knn_f64_dev(const double* dX, int64_t nq, const double* dY, int64_t nr, int32_t d, int32_t K, int32_t self_mode,
                           int64_t self_offset, const double* d_dist, int32_t ld, int32_t nsample, uint64_t seed, int32_t* d_result,
                           void* ws, size_t ws_bytes, void* stream);
int mce_verify_knn_f64(const double* X, int64_t nq, const double* Y, int64_t nr, int32_t d, int32_t K, int32_t self_mode,
                       int64_t self_offset, const double* dist, int32_t ld, int32_t nsample, uint64_t seed, int32_t* failed,
                       int32_t device);

/* Spatial pruning of the fp16-filter search for low-dimensional, large reference sets (d <= 13):
 * both point sets are put in k-d order on the device (cells of 32 rows) and every wave of 64 queries
 * visits the reference chunks nearest-box-first, multiplies only the 32-row tiles whose box is within
 * reach, and stops once no remaining chunk can hold a neighbour.  Same neighbours, distances and
 * tie-breaks as the exhaustive search.  0 (default): used where it was measured faster -- d <= 4 from
 * 100 k reference rows, d = 5 from 125 k, d = 6 from 150 k, d = 7 from 250 k, d = 8 from 500 k (the table is capi_common.hpp: kPruneAutoMinRows),
 * and at least 32 k queries, no fewer than an eighth of the reference rows; 1: never; 2: whenever the shape allows it
 * (d <= 15, K <= 16).  Process-wide default (per call: mce_options). */
int mce_set_prune_mode(int mode);
int mce_get_prune_mode(void);

/* Symmetric sweep of an auto-evidence search (X and Y are ONE buffer, nq == nr: reference
 * MCEvidence.py:1100-1104, `nbrs.fit(samples); nbrs.kneighbors(samples)`).  d(i,j) = d(j,i), so every
 * 32x32 tile of the exhaustive fp16 filter sweep is needed by both of its sides; here it is multiplied once:
 * the rows are sorted by distance from the mean, a prepass bounds every row's K-th distance, every block of
 * 512 rows sweeps only the blocks before it, and each tile is gated for the streamed rows
 * too; their candidates are merged into the lists afterwards.  Same neighbours, distances and tie-breaks as
 * the exhaustive search.  0 (default): where it was measured faster and pruning does not apply -- from 768 blocks of
 * 512 rows (393 k rows; twice that with K > 12) where the filter takes one 16-wide k-step (d <= 15: capi.hip
 * f16_ksteps(d) = (d + 16) / 16), from 257 blocks (131 k rows) with two (d <= 31), from 193 (99 k rows) beyond (kSymAutoMinBlocks);
 * 1: never; 2: whenever the shape allows it (fp16-filter shapes with K <= 16, more than 512 rows).
 * Process-wide default (per call: mce_options); the environment variable MCE_SYM sets the initial value.
 * Multi-GPU: up to four ranks mce_knn_dotp_part_f64 partitions such a search by ranges of the sorted blocks (symmetric
 * within a rank's range, column side only against the other ranks' rows) -- no exchange between the ranks; larger jobs
 * shard the query rows. */
int mce_set_sym_mode(int mode);
int mce_get_sym_mode(void);
/* Work actually done by the last pruned search launched by this thread through a *_dev entry point
 * (its workspace must still be alive): fraction of (query block, reference chunk) pairs staged, and
 * fraction of (wave, 32-row tile) products multiplied.  Synchronises the device. */
int mce_last_prune_stats(double *chunk_fraction, double *tile_fraction);

/* Measurement hook: while enabled, the search-kernel launch of every call on this thread
 * is bracketed by hipEvents recorded on the launch stream (no synchronisation);
 * mce_last_kernel_ms() waits for the brackets recorded since the last enable and returns
 * the MEAN search-kernel time per call in ms (-1 if none; a call that searches two query ranges --
 * DESIGN.md 3.0, trailing round -- counts once, with both launches).  Used by bench.py for the
 * roofline figure. */
void mce_set_profiling(int on);
double mce_last_kernel_ms(void);
/* What the last search on this thread asked of the matrix cores, for rooflines that follow from a profile (bench.py):
 *   out[0]  MFMA flops EXECUTED by the dominant kernel (the one mce_last_kernel_ms() brackets) -- a symmetric sweep
 *           multiplies each pair of rows once, so this is about half of the all-pairs figure; -1: only device counters
 *           know (pruned walk: mce_last_prune_stats)
 *   out[1]  the same over every launch of the search (prepass / seed phases included)
 *   out[2]  milliseconds of the whole search, packing to the last list kernel (HIP events on the launch stream, mean per
 *           search since mce_set_profiling(1)); -1 without profiling
 *   out[3]  mce_last_kernel_ms()
 * Bookkeeping of the measurement hooks; no counterpart in the reference. */
int mce_last_search_stats(double* out, int32_t n);

/* Test hook: one 32 x 32 tile of the filter's matrix product exactly as the search kernels issue it
 * (v_mfma_f32_32x32x16_f16, `kst` chained k-steps): out[row * 32 + query] = sum_k yprime[row][k] * xprime[query][k], rows of
 * 16 * kst fp16 values (bit patterns) each.  tests/test_gpu_parity.py::test_mfma_error_model checks the error model the
 * rigorous filter bound assumes (knn_f16.hpp) against it.  No counterpart in the reference. */
int mce_debug_mfma_tile_f16(const uint16_t* yprime, const uint16_t* xprime, int32_t kst, float* out, int32_t device);
/* the same for `ntiles` independent tiles in one launch (yprime, xprime: [ntiles][32][16 kst]; out: [ntiles][32][32]) */
int mce_debug_mfma_tiles_f16(const uint16_t* yprime, const uint16_t* xprime, int32_t kst, int32_t ntiles, float* out, int32_t device);

#ifdef __cplusplus
}
#endif
#endif /* MCEVIDENCE_HIP_H */
